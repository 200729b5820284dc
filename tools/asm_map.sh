# tools/asm_map.sh <kernel-name-regex> [nth]: loads / waits / barriers / MFMA counts in the loops of a kernel of ttrnn_fast_c2w.hip
cd /root/repo/tensorized-rnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function --cuda-device-only -S ttrnn_fast_c2w.hip -o /tmp/c2w_dev.s 2>&1 | grep -v hip-link | head
N=${2:-1}
A=$(grep -n "^_ZN.*$1.*:" /tmp/c2w_dev.s | sed -n "${N}p" | cut -d: -f1)
B=$(awk -v a=$A 'NR>a && /^\.Lfunc_end/{print NR; exit}' /tmp/c2w_dev.s)
sed -n "${A},${B}p" /tmp/c2w_dev.s | awk '/Loop Header/{f=1} f{ if ($0 ~ /v_mfma/) m++; if ($0 ~ /global_load/) {print NR": load (mfma so far "m")"}; if ($0 ~ /s_barrier/) {print NR": barrier (mfma "m")"}; if ($0 ~ /s_waitcnt vmcnt/) print NR": "$0" (mfma "m")"; if ($0 ~ /s_cbranch/ ) {print NR": "$0} }'
