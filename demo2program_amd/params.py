"""Parameter inventory, initialisers and the flat device buffers.

All trainable tensors live in ONE flat fp32 device buffer (each tensor 16-byte aligned);
gradients, Adam m and v are flat buffers of the same layout.  The data-parallel all-reduce,
the global-norm clip and the Adam update each run over the whole buffer in a single call
(SURVEY.md 8(a) A13, 8(e)).  Names follow the reference's variable scopes
(models/model_full.py:216-316,497-599).
"""
from collections import OrderedDict

import numpy as np

from .config import conv_shapes, feature_dim


def param_shapes(config):
    U = config.num_lstm_cell_units
    s = OrderedDict()
    for l, (_, _, cin, cout, _, _) in enumerate(conv_shapes(config), start=1):
        s['conv%d/W' % l] = (3, 3, cin, cout)
        s['conv%d/b' % l] = (cout,)
        s['conv%d/beta' % l] = (cout,)
        s['conv%d/gamma' % l] = (cout,)
    s['demo_lstm/kernel'] = (feature_dim(config) + U, 4 * U)
    s['demo_lstm/bias'] = (4 * U,)
    s['second_lstm/kernel'] = (2 * U, 4 * U)
    s['second_lstm/bias'] = (4 * U,)
    for sc in ('rn_h', 'rn_c'):
        s[sc + '/fc1/W'] = (2 * U, U)
        s[sc + '/fc1/b'] = (U,)
        s[sc + '/fc1/beta'] = (U,)
        s[sc + '/fc1/gamma'] = (U,)
        s[sc + '/fc2/W'] = (U, U)
        s[sc + '/fc2/b'] = (U,)
        s[sc + '/fc2/beta'] = (U,)
        s[sc + '/fc2/gamma'] = (U,)
    s['prog/embedding'] = (config.dim_program_token + 1, U)
    s['prog/lstm/kernel'] = (2 * U, 4 * U)
    s['prog/lstm/bias'] = (4 * U,)
    s['prog/proj'] = (U, config.dim_program_token)
    s['act/embedding'] = (config.action_space + 1, U)
    s['act/lstm/kernel'] = (2 * U, 4 * U)
    s['act/lstm/bias'] = (4 * U,)
    s['act/proj'] = (U, config.action_space)
    s['per/fc/W'] = (config.per_dim, U)
    s['per/fc/b'] = (U,)
    s['per/fc/beta'] = (U,)
    s['per/fc/gamma'] = (U,)
    s['per/lstm/kernel'] = (2 * U, 4 * U)
    s['per/lstm/bias'] = (4 * U,)
    s['per/proj'] = (U, config.per_dim)
    model = getattr(config, 'model', 'full')
    if model == 'summarizer':            # models/baselines/model_summarizer.py: no multi-task decoders
        drop = ('act/', 'per/')
    elif model == 'synthesis_baseline':  # model_synthesis.py: nor second pass / relation network
        drop = ('act/', 'per/', 'second_lstm/', 'rn_h/', 'rn_c/')
    else:
        drop = None
    if drop:
        s = OrderedDict((n, sh) for n, sh in s.items() if not n.startswith(drop))
    return s


def num_params(config):
    return int(sum(int(np.prod(sh)) for sh in param_shapes(config).values()))


def init_params(config, seed=123):
    """[TF-1.3] default initialisers of the reference's layers:
    slim conv / fc: xavier uniform +-sqrt(6/(fan_in+fan_out)) (conv fan = 3*3*C);
    LSTM kernels and Dense: glorot uniform; biases 0; BN beta 0 / gamma 1;
    embeddings U(-0.01, 0.01) (models/model_full.py:288-291).
    Values come from numpy's RandomState(seed), not TF's stream: parity is defined on given
    weights, never on the initialiser."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, sh in param_shapes(config).items():
        leaf = name.split('/')[-1]
        if leaf in ('b', 'bias', 'beta'):
            v = np.zeros(sh, np.float32)
        elif leaf == 'gamma':
            v = np.ones(sh, np.float32)
        elif leaf == 'embedding':
            v = rs.uniform(-0.01, 0.01, sh).astype(np.float32)
        else:
            if len(sh) == 4:
                fan_in, fan_out = sh[0] * sh[1] * sh[2], sh[0] * sh[1] * sh[3]
            else:
                fan_in, fan_out = sh[0], sh[1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            v = rs.uniform(-lim, lim, sh).astype(np.float32)
        out[name] = v
    return out


class FlatParams(object):
    """Flat device buffers + named views.  ``p[name]``, ``g[name]`` are tensor views."""

    ALIGN = 4   # floats (16 bytes): every kernel may use dwordx4 accesses

    def __init__(self, config, values=None, seed=123, device='cuda'):
        import torch
        self.shapes = param_shapes(config)
        self.offsets = OrderedDict()
        off = 0
        for name, sh in self.shapes.items():
            self.offsets[name] = off
            n = int(np.prod(sh))
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.size = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        # the gradient buffer carries ONE spare 16-byte slot behind the parameters' gradients: the step-status word
        # of the guarded optimizer step travels through the data-parallel all-reduce in it (`grad_all` is what the
        # exchange step reduces; `grad` -- the norm, the clip, Adam -- never sees the slot)
        self.grad_all = torch.zeros(off + self.ALIGN, dtype=torch.float32, device=device)
        self.grad = self.grad_all[:off]
        self.status_slot = self.grad_all[off:]
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.p = OrderedDict()
        self.g = OrderedDict()
        for name, sh in self.shapes.items():
            o, n = self.offsets[name], int(np.prod(sh))
            self.p[name] = self.flat[o:o + n].view(*sh)
            self.g[name] = self.grad[o:o + n].view(*sh)
        self.load(values if values is not None else init_params(config, seed))

    def load(self, values):
        import torch
        host = np.zeros(self.size, np.float32)
        for name, sh in self.shapes.items():
            v = np.asarray(values[name], dtype=np.float32)
            assert tuple(v.shape) == tuple(sh), (name, v.shape, sh)
            o = self.offsets[name]
            host[o:o + v.size] = v.reshape(-1)
        self.flat.copy_(torch.from_numpy(host))

    def to_numpy(self, which='p'):
        src = {'p': self.flat, 'g': self.grad, 'm': self.m, 'v': self.v}[which].cpu().numpy()
        out = OrderedDict()
        for name, sh in self.shapes.items():
            o, n = self.offsets[name], int(np.prod(sh))
            out[name] = src[o:o + n].reshape(sh).copy()
        return out
