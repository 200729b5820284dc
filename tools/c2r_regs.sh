# registers / scratch of every k_c2r instantiation
cd /root/repo/tensorized-rnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -c ttrnn_fast_c2w.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning|Function Name.*k_c2rI|    VGPRs:|ScratchSize" | grep -A2 "k_c2rI\|error\|warning" | cut -c1-200 | grep -v "^--" | paste - - - | sed 's/.*C2RSI\([^ ]*\)E.*VGPRs: \([0-9]*\).*ScratchSize \[bytes\/lane\]: \([0-9]*\).*/\1 vgpr \2 scratch \3/' | cut -c1-150
