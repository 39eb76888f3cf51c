"""ConvTemporalGraphical: the graph-convolution unit, running on the fused gfx950 kernel.

Drop-in for the reference module of the same name (net/utils/tgcn.py:6-66 of hensyou/st-gcn):
same constructor, same `conv` sub-module (so state-dict keys `conv.weight` / `conv.bias` and the
class-name based weights_init of processor/recognition.py:21-33 keep working), same
forward(x, A) -> (y, A) contract with x [N, C_in, T, V] and A [K, V, V].  `self.conv` only
holds the parameters: its own forward is never called.  With t_kernel_size=1 (every st_gcn block) the
projection and the aggregation run as one HIP kernel and the K*C intermediate is never materialised;
other temporal kernels, strides, paddings and dilations run as a temporal conv kernel plus that kernel.
"""
import torch
import torch.nn as nn

from stgcn_rt import ops as P
from stgcn_rt.functional import GraphConvFn, TemporalConvFn


class ConvTemporalGraphical(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1,
                 t_padding=0, t_dilation=1, bias=True):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, kernel_size=(t_kernel_size, 1),
                              padding=(t_padding, 0), stride=(t_stride, 1), dilation=(t_dilation, 1), bias=bias)
        self.t_cfg = (t_kernel_size, t_stride, t_padding, t_dilation)
        self.compute_dtype = None          # None: follow the input's dtype (float32 or bfloat16)

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size
        dtype = self.compute_dtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        cin = self.conv.in_channels
        if self.t_cfg != (1, 1, 0, 1):
            return self._forward_temporal(x, A, dtype), A
        xin = P.pad_channels(x.to(dtype), P.pad_to(cin, P.kstep(dtype)))
        y = GraphConvFn.apply(xin, A, self.conv.weight, self.conv.bias, cin)
        return y, A

    def _forward_temporal(self, x, A, dtype):
        """t_kernel_size > 1, stride, padding, dilation (net/utils/tgcn.py:48-66; no st_gcn block uses them): the
        temporal convolution to K*C channels runs on the implicit-GEMM conv kernel, the aggregation on the fused
        graph-conv kernel with a selection matrix as its projection (row k*C+c picks channel k*C+c -- exact in either
        precision), so both directions stay on the MFMA kernels at the cost of one K*C-wide intermediate."""
        K = self.kernel_size
        cin = self.conv.in_channels
        kc = self.conv.out_channels                     # K * C
        xin = P.pad_channels(x.to(dtype), P.pad_to(cin, P.kchunk(dtype)))
        z = TemporalConvFn.apply(xin, self.conv.weight, self.conv.bias, self.t_cfg, cin)
        kc_p = z.shape[1]
        sel = getattr(self, "_sel", None)
        if sel is None or sel.shape[1] != kc_p or sel.device != z.device:
            sel = torch.zeros((kc, kc_p, 1, 1), dtype=torch.float32, device=z.device)
            sel[torch.arange(kc), torch.arange(kc)] = 1.0
            self._sel = sel
        return GraphConvFn.apply(z, A, sel, None, kc_p)
