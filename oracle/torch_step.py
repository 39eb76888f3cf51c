"""The ST-GCN training step on stock PyTorch CPU operators.  TEST INFRASTRUCTURE ONLY (see stgcn_oracle.py).

A second, independent restatement of the hot path: where stgcn_oracle.py spells every forward and backward formula out,
this one calls the library operators the reference itself delegates to -- F.conv2d (net/utils/tgcn.py:48-60,
net/st_gcn.py:162-168, 181-185), torch.einsum (tgcn.py:64), F.batch_norm (st_gcn.py:42, 160, 169, 186), F.dropout
(st_gcn.py:170), autograd, torch.optim.SGD (processor/recognition.py:46-53) -- as a function of a flat state dict
with the reference's names.  It is what bench.py times as the CPU baseline (the reference's CPU path is these same
operators), and tests/test_oracle_golden.py pins it to the golden vectors like the explicit oracle.
"""
import torch
import torch.nn.functional as F

from . import stgcn_oracle as O


def _bn(x, sd, pre, training):
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], sd[pre + ".weight"], sd[pre + ".bias"],
                        training=training, momentum=0.1, eps=1e-5)


def forward(sd, x, training, dropout=0.0):
    """x [N,C,T,V,M] -> logits [N,num_class]; running statistics in sd are updated in place when training."""
    N, C, T, V, M = x.shape
    A = sd["A"]
    K = A.shape[0]
    h = _bn(x.permute(0, 4, 3, 1, 2).reshape(N * M, V * C, T), sd, "data_bn", training)
    h = h.reshape(N, M, V, C, T).permute(0, 1, 3, 4, 2).reshape(N * M, C, T, V)
    for i, (cin, cout, stride, res) in enumerate(O.BLOCK_SPECS):
        cin = C if cin is None else cin
        pre = "st_gcn_networks.%d." % i
        Ai = A * sd["edge_importance.%d" % i] if ("edge_importance.%d" % i) in sd else A
        kind = O.block_residual_kind(cin, cout, stride, res)
        if kind == "identity":
            r = h
        elif kind == "conv":
            r = _bn(F.conv2d(h, sd[pre + "residual.0.weight"], sd[pre + "residual.0.bias"], stride=(stride, 1)), sd, pre + "residual.1", training)
        else:
            r = 0
        z = F.conv2d(h, sd[pre + "gcn.conv.weight"], sd[pre + "gcn.conv.bias"])
        y = torch.einsum("nkctv,kvw->nctw", z.reshape(z.shape[0], K, cout, z.shape[2], V), Ai)
        q = F.relu(_bn(y, sd, pre + "tcn.0", training))
        q = _bn(F.conv2d(q, sd[pre + "tcn.2.weight"], sd[pre + "tcn.2.bias"], stride=(stride, 1), padding=(4, 0)), sd, pre + "tcn.3", training)
        if i > 0 and dropout > 0:
            q = F.dropout(q, dropout, training)
        h = F.relu(q + r)
    pooled = h.mean(dim=(2, 3)).reshape(N, M, -1).mean(dim=1)
    return F.conv2d(pooled[:, :, None, None], sd["fcn.weight"], sd["fcn.bias"]).reshape(N, -1)


class Stepper:
    """REC_Processor.train's iteration (processor/recognition.py:84-97) on a state dict: forward, mean
    cross-entropy, backward, SGD-nesterov with weight decay on every parameter."""

    def __init__(self, sd, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True, dropout=0.0):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = O.param_names(self.sd)
        for k in self.names:
            self.sd[k].requires_grad_(True)
        self.dropout = dropout
        self.opt = torch.optim.SGD([self.sd[k] for k in self.names], lr=lr, momentum=momentum, nesterov=nesterov,
                                   weight_decay=weight_decay)

    def step(self, x, labels):
        self.opt.zero_grad()
        loss = F.cross_entropy(forward(self.sd, x, True, self.dropout), labels)
        loss.backward()
        self.opt.step()
        return loss.detach()
