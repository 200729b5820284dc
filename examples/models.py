"""Callers of the hot path, restated for the drop-in packages (harness rows H1-H3).

`MNISTClassifier` has the shape contract of the reference's experiments/digit_classification/
mnist_classifier.py:13-57 (TT-RNN -> TTLinear head on the last timestep -> log_softmax);
`SpeakerEncoder.forward` that of experiments/speaker_verification/encoder/speaker_encoder.py:69-91
(TT-RNN -> TTLinear on the last hidden state -> ReLU -> L2 normalisation); its GE2E similarity matrix / loss
(:93-170) are the vectorised device-side versions of ttrnn_hip/ge2e.py.  Both are ordinary nn.Modules over
`tensorized_rnn` / `t3nsor` from tensorized-rnn_amd/, so every matmul-shaped op runs in libttrnn.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch import nn  # noqa: E402

from t3nsor.layers import TTLinear  # noqa: E402
from tensorized_rnn.gru import TTGRU  # noqa: E402
from tensorized_rnn.tt_lstm import TTLSTM  # noqa: E402


class MNISTClassifier(nn.Module):
    def __init__(self, input_size, output_size, hidden_size, num_layers, device, gru=False, n_cores=3, tt_rank=8,
                 naive_tt=False, extra_core=None, log_grads=False):
        super().__init__()
        self.gru = gru
        cls = TTGRU if gru else TTLSTM
        self.rnn = cls(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, device=device,
                       n_cores=n_cores, tt_rank=tt_rank, log_grads=log_grads, is_naive=naive_tt, new_core=extra_core)
        self.linear = TTLinear(in_features=hidden_size, out_features=output_size, bias=True, auto_shapes=True,
                               d=n_cores, tt_rank=tt_rank)

    def param_count(self):
        return self.rnn.param_count() + sum(p.numel() for p in self.linear.parameters())

    def forward(self, inputs):
        # the reference classifies outputs[:, -1, :] (mnist_classifier.py:52-55), which IS the last layer's final hidden state:
        # taking it from the state instead of slicing the [B, T, H] outputs spares autograd a zero-filled [B, T, H] gradient
        # (51 MB per cfg2 step) that the reverse-time kernel would then read
        # (inference: the [B, T, H] outputs of the last layer are not even written — need_outputs=False)
        res = self.rnn(inputs) if torch.is_grad_enabled() else self.rnn(inputs, need_outputs=False)
        last = res[1] if self.gru else res[1][0]
        return self.linear.forward_head(last, "log_softmax")                # TTLinear + log_softmax: one library call


class SpeakerEncoder(nn.Module):
    def __init__(self, mel_n_channels, hidden_size, num_layers, embedding_size, device, n_cores=3, rank=16,
                 use_gru=False):
        super().__init__()
        cls = TTGRU if use_gru else TTLSTM
        self.use_gru = use_gru
        self.rnn = cls(mel_n_channels, hidden_size, num_layers, device, n_cores=n_cores, tt_rank=rank)
        self.linear = TTLinear(in_features=hidden_size, out_features=embedding_size, bias=True, auto_shapes=True,
                               d=n_cores, tt_rank=rank).to(device)
        self.similarity_weight = nn.Parameter(torch.tensor([10.]))
        self.similarity_bias = nn.Parameter(torch.tensor([-5.]))

    def forward(self, utterances):
        # only the last layer's final hidden state is consumed (speaker_encoder.py:80-86): inference skips its [B, T, H] outputs
        res = self.rnn(utterances) if torch.is_grad_enabled() else self.rnn(utterances, need_outputs=False)
        last_hidden = res[1] if self.use_gru else res[1][0]
        return self.linear.forward_head(last_hidden, "relu_l2norm")              # TTLinear + ReLU + L2 norm: one library call

    def similarity_matrix(self, verification_embeds, enrollment_embeds=None):
        """speaker_encoder.py:93-140, vectorised and on the embeddings' device (ttrnn_hip/ge2e.py)."""
        from ttrnn_hip import ge2e
        return ge2e.similarity_matrix(verification_embeds, self.similarity_weight.to(verification_embeds.device),
                                      self.similarity_bias.to(verification_embeds.device), enrollment_embeds)

    def loss(self, verification_embeds, enrollment_embeds=None):
        """(loss, eer) as speaker_encoder.py:142-170; under torch.distributed every rank passes the [S_local, U, D]
        embeddings of its own speakers (ge2e.ge2e_loss_data_parallel: one all-gather)."""
        from ttrnn_hip import ge2e
        w = self.similarity_weight.to(verification_embeds.device)
        b = self.similarity_bias.to(verification_embeds.device)
        if enrollment_embeds is None and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            return ge2e.ge2e_loss_data_parallel(verification_embeds, w, b)
        return ge2e.ge2e_loss(verification_embeds, w, b, enrollment_embeds)

    def do_gradient_ops(self, clip=3.0):
        """speaker_encoder.py:60-66: scale the similarity parameters' gradients by 0.01, clip the global norm."""
        self.similarity_weight.grad *= 0.01
        self.similarity_bias.grad *= 0.01
        torch.nn.utils.clip_grad_norm_(self.parameters(), clip, norm_type=2)
