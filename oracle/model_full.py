"""Torch-CPU restatement of the reference full-model graph, loss and optimizer.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED against
TF-1.3 (no TensorFlow available); pinned by known-answer tests.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  The structure deliberately mirrors the reference --
the Demo encoder, SecondPath encoder and the action / perception decoders are
invoked ``k`` times in a Python loop, one call per demonstration index, exactly
as ``models/model_full.py:373-398,530-599`` does -- so that per-call batch-norm
statistics fall out of the structure instead of being re-derived.
The HIP product path batches those k calls; agreement between the two is what
the parity tests check.

All tensors are torch CPU tensors; ``dtype`` is float32 for parity/timing and
float64 when a tighter reference is wanted.  Gradients come from torch autograd.
"""
from dataclasses import dataclass
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config

@dataclass
class OracleConfig:
    """Data dimensions the reference derives at trainer.py:312-321."""
    batch_size: int
    k: int
    max_demo_len: int
    max_program_len: int
    h: int
    w: int
    depth: int
    dim_program_token: int
    action_space: int
    per_dim: int
    num_lstm_cell_units: int = 512
    dataset_type: str = 'karel'
    # 'full' (models/model_full.py) or one of the program-synthesis ablations
    # 'summarizer' (models/baselines/model_summarizer.py), 'synthesis_baseline'
    # (models/baselines/model_synthesis.py); demo_aggregation only matters for the latter
    model: str = 'full'
    demo_aggregation: str = 'avgpool'

    @property
    def n_conv(self):
        # models/model_full.py:219-229 -- 3 convs, +2 for vizdoom
        return 5 if self.dataset_type == 'vizdoom' else 3

    @property
    def conv_channels(self):
        return [16, 32, 48, 48, 48][: self.n_conv]

    @property
    def feature_dim(self):
        h, w = self.h, self.w
        for _ in range(self.n_conv):
            h, w = (h + 1) // 2, (w + 1) // 2
        return h * w * self.conv_channels[-1]


def param_shapes(cfg):
    """name -> shape, in PARAM_ORDER order."""
    U = cfg.num_lstm_cell_units
    shapes = {}
    cin = cfg.depth
    for l, cout in enumerate(cfg.conv_channels, start=1):
        shapes['conv%d/W' % l] = (3, 3, cin, cout)
        shapes['conv%d/b' % l] = (cout,)
        shapes['conv%d/beta' % l] = (cout,)
        shapes['conv%d/gamma' % l] = (cout,)
        cin = cout
    shapes['demo_lstm/kernel'] = (cfg.feature_dim + U, 4 * U)
    shapes['demo_lstm/bias'] = (4 * U,)
    shapes['second_lstm/kernel'] = (2 * U, 4 * U)
    shapes['second_lstm/bias'] = (4 * U,)
    for s in ('rn_h', 'rn_c'):
        shapes[s + '/fc1/W'] = (2 * U, U)
        shapes[s + '/fc1/b'] = (U,)
        shapes[s + '/fc1/beta'] = (U,)
        shapes[s + '/fc1/gamma'] = (U,)
        shapes[s + '/fc2/W'] = (U, U)
        shapes[s + '/fc2/b'] = (U,)
        shapes[s + '/fc2/beta'] = (U,)
        shapes[s + '/fc2/gamma'] = (U,)
    shapes['prog/embedding'] = (cfg.dim_program_token + 1, U)
    shapes['prog/lstm/kernel'] = (2 * U, 4 * U)
    shapes['prog/lstm/bias'] = (4 * U,)
    shapes['prog/proj'] = (U, cfg.dim_program_token)
    shapes['act/embedding'] = (cfg.action_space + 1, U)
    shapes['act/lstm/kernel'] = (2 * U, 4 * U)
    shapes['act/lstm/bias'] = (4 * U,)
    shapes['act/proj'] = (U, cfg.action_space)
    shapes['per/fc/W'] = (cfg.per_dim, U)
    shapes['per/fc/b'] = (U,)
    shapes['per/fc/beta'] = (U,)
    shapes['per/fc/gamma'] = (U,)
    shapes['per/lstm/kernel'] = (2 * U, 4 * U)
    shapes['per/lstm/bias'] = (4 * U,)
    shapes['per/proj'] = (U, cfg.per_dim)
    model = getattr(cfg, 'model', 'full')
    if model != 'full':
        # the baselines have no multi-task decoders; the synthesis baseline also has neither the
        # second encoder pass nor the relation network (model_synthesis.py:324-358)
        drop = ('act/', 'per/') if model == 'summarizer' else ('act/', 'per/', 'second_lstm/', 'rn_h/', 'rn_c/')
        shapes = {n: sh for n, sh in shapes.items() if not n.startswith(drop)}
    return shapes


def PARAM_ORDER(cfg):
    return list(param_shapes(cfg).keys())


# --------------------------------------------------------------------------- layer ops

BN_EPS = 1e-3     # [TF-1.3] contrib.layers.batch_norm default epsilon=0.001
BN_DECAY = 0.9    # models/ops.py:21


def lrelu(x, leak=0.2):
    """models/ops.py:7-11: f1*x + f2*abs(x), f1=.5(1+leak), f2=.5(1-leak)."""
    f1 = 0.5 * (1 + leak)
    f2 = 0.5 * (1 - leak)
    return f1 * x + f2 * x.abs()


def same_pad_s2k3(n):
    """[TF-1.3] SAME padding for kernel 3, stride 2.

    out = ceil(n/2); total = max((out-1)*2 + 3 - n, 0); before = total//2.
    Even n -> (0, 1); n = 5 -> (1, 1); n = 1 -> (1, 1)."""
    out = (n + 1) // 2
    total = max((out - 1) * 2 + 3 - n, 0)
    before = total // 2
    return before, total - before


def batch_norm_train(x, beta, gamma):
    """[TF-1.3] contrib.layers.batch_norm(is_training=True), non-fused
    (models/ops.py:20-23): per-channel (last axis) mean and BIASED variance over
    all other axes of this call, eps 1e-3.  Returns (y, mean, var)."""
    red = tuple(range(x.dim() - 1))
    mean = x.mean(dim=red)
    var = ((x - mean) ** 2).mean(dim=red)
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
    return y, mean, var


def batch_norm_infer(x, beta, gamma, moving_mean, moving_var):
    """[TF-1.3] contrib.layers.batch_norm(is_training=False) (models/ops.py:20-23 with
    is_train=False, as evaler.py:61 builds the model): normalise with the moving statistics."""
    return (x - moving_mean) * torch.rsqrt(moving_var + BN_EPS) * gamma + beta


def _bn(x, beta, gamma, moving=None):
    """Training-mode BN, or inference BN when `moving` = (moving_mean, moving_var) is given."""
    if moving is None:
        return batch_norm_train(x, beta, gamma)
    mm, mv = moving
    return batch_norm_infer(x, beta, gamma, mm.to(x.dtype), mv.to(x.dtype)), mm, mv


def conv2d_lrelu_bn(x, W, b, beta, gamma, moving=None):
    """models/ops.py:27-33 as called from State_Encoder (model_full.py:219-229):
    slim.conv2d 3x3 stride 2 SAME + bias -> lrelu(0.2) -> BN(train).

    x: [N, H, W, C] (NHWC); W: [3, 3, Cin, Cout] (TF HWIO)."""
    N, H, Wd, C = x.shape
    pt, pb = same_pad_s2k3(H)
    pl, pr = same_pad_s2k3(Wd)
    xn = x.permute(0, 3, 1, 2)                       # NCHW for torch
    xn = F.pad(xn, (pl, pr, pt, pb))
    wt = W.permute(3, 2, 0, 1)                       # OIHW
    y = F.conv2d(xn, wt, bias=b, stride=2)
    y = y.permute(0, 2, 3, 1)                        # back to NHWC
    a = lrelu(y)
    out, mean, var = _bn(a, beta, gamma, moving)
    return out, mean, var


def fc_lrelu_bn(x, W, b, beta, gamma, act=True, moving=None):
    """models/ops.py:149-155: slim.fully_connected (+bias) -> [lrelu] -> BN.
    Contracts the last axis; BN over all other axes."""
    y = x @ W + b
    if act:
        y = lrelu(y)
    return _bn(y, beta, gamma, moving)


def basic_lstm_cell(x, c, h, kernel, bias, forget_bias=1.0):
    """[TF-1.3] rnn.BasicLSTMCell.call: concat([x,h])·W + b, split
    into i, j, f, o; c' = c·σ(f+1) + σ(i)·tanh(j); h' = tanh(c')·σ(o)."""
    z = torch.cat([x, h], dim=1) @ kernel + bias
    i, j, f, o = torch.chunk(z, 4, dim=1)
    c_new = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
    h_new = torch.tanh(c_new) * torch.sigmoid(o)
    return c_new, h_new


def dynamic_rnn(x, seq_len, kernel, bias, c0=None, h0=None):
    """[TF-1.3] tf.nn.dynamic_rnn(BasicLSTMCell, sequence_length=seq_len)
    (model_full.py:254-256,274-276).  For t >= len: output row is 0,
    state copied through.  x: [B, T, I].  Returns (outputs [B,T,U], h, c)."""
    B, T, _ = x.shape
    U = kernel.shape[1] // 4
    c = torch.zeros(B, U, dtype=x.dtype) if c0 is None else c0
    h = torch.zeros(B, U, dtype=x.dtype) if h0 is None else h0
    outs = []
    for t in range(T):
        c_new, h_new = basic_lstm_cell(x[:, t], c, h, kernel, bias)
        active = (t < seq_len).to(x.dtype).unsqueeze(1)
        outs.append(h_new * active)
        c = c_new * active + c * (1 - active)
        h = h_new * active + h * (1 - active)
    return torch.stack(outs, dim=1), h, c


def embedding_lookup_oob0(table, ids):
    """[TF-1.3] tf.nn.embedding_lookup on GPU: out-of-range id -> zero row, no
    gradient (model_full.py:294,448-450)."""
    ok = (ids >= 0) & (ids < table.shape[0])
    safe = torch.where(ok, ids, torch.zeros_like(ids))
    out = table[safe]
    return out * ok.unsqueeze(-1).to(table.dtype)


def training_decoder(inputs, seq_len, c0, h0, kernel, bias, proj, max_len):
    """[TF-1.3] BasicDecoder(cell, TrainingHelper(inputs, seq_len), (c0,h0),
    Dense(no bias)) under dynamic_decode(impute_finished=False,
    maximum_iterations=max_len) (model_full.py:413,463-489).

    Steps run until every row is finished: n_steps = max(seq_len) (capped by
    max_len); rows past their own length keep computing; afterwards logits are
    zero-padded to max_len and transposed to [B, token_dim, max_len]."""
    B = inputs.shape[0]
    n_steps = int(min(int(seq_len.max().item()), max_len)) if B > 0 else 0
    c, h = c0, h0
    logits = []
    for t in range(n_steps):
        c, h = basic_lstm_cell(inputs[:, t], c, h, kernel, bias)
        logits.append(h @ proj)
    token_dim = proj.shape[1]
    if n_steps > 0:
        out = torch.stack(logits, dim=1)
    else:
        out = torch.zeros(B, 0, token_dim, dtype=inputs.dtype)
    pad = torch.zeros(B, max_len - n_steps, token_dim, dtype=inputs.dtype)
    out = torch.cat([out, pad], dim=1)
    return out.permute(0, 2, 1)                      # [B, token_dim, max_len]


def sigmoid_xent(logits, labels):
    """[TF-1.3] sigmoid_cross_entropy_with_logits:
    max(x,0) - x*z + log(1 + exp(-|x|))."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


def softmax_xent(logits, labels):
    """[TF-1.3] softmax_cross_entropy_with_logits: -sum(labels*log_softmax)
    (all-zero label row -> 0)."""
    return -(labels * F.log_softmax(logits, dim=-1)).sum(dim=-1)


def sequence_loss(pred, gt, gt_len, max_len, token_dim, sequence_type):
    """model_full.py:620-657: masked, mask-count-normalised sequence loss.

    pred, gt: [B, token_dim, max_len]; gt_len: [B] int."""
    B = pred.shape[0]
    pos = torch.arange(max_len).unsqueeze(0)
    gt_mask = (pos < gt_len.unsqueeze(1)).to(pred.dtype)          # tf.sequence_mask
    labels = gt.permute(0, 2, 1).reshape(B * max_len, token_dim)
    logits = pred.permute(0, 2, 1).reshape(B * max_len, token_dim)
    if sequence_type in ('program', 'action'):
        ce = softmax_xent(logits, labels)
    else:
        ce = sigmoid_xent(logits, labels).mean(dim=-1)
    return (ce * gt_mask.reshape(-1)).sum() / gt_mask.sum()


def rn_pool(feat, p, scope, moving=None):
    """model_full.py:333-349: relation network over ordered demo pairs.
    feat: [B, k, U]."""
    B, k, U = feat.shape
    tile1 = feat.unsqueeze(1).expand(B, k, k, U)     # tile1[b,a,c] = feat[b,c]
    tile2 = feat.unsqueeze(2).expand(B, k, k, U)     # tile2[b,a,c] = feat[b,a]
    x = torch.cat([tile1, tile2], dim=3).reshape(B * k * k, 2 * U)
    mv = (lambda n: None) if moving is None else (lambda n: moving[scope + n])
    x, _, _ = fc_lrelu_bn(x, p[scope + '/fc1/W'], p[scope + '/fc1/b'],
                          p[scope + '/fc1/beta'], p[scope + '/fc1/gamma'], moving=mv('/fc1'))
    x, _, _ = fc_lrelu_bn(x, p[scope + '/fc2/W'], p[scope + '/fc2/b'],
                          p[scope + '/fc2/beta'], p[scope + '/fc2/gamma'], moving=mv('/fc2'))
    return x.reshape(B, k, k, U).mean(dim=1).mean(dim=1)


# --------------------------------------------------------------------------- the graph

def forward(p, batch, cfg, moving=None, fed_ids=None):
    """models/model_full.py:208-600 (graph) + :918-932,1014-1038,1061-1079 (loss).
    moving: None = training-mode BN (batch statistics per call); a dict name -> (moving_mean,
    moving_var) for 'conv<l>', 'rn_h/fc1', 'rn_h/fc2', 'rn_c/fc1', 'rn_c/fc2', 'per/fc' = the
    is_train=False graph of evaler.py:61.
    fed_ids: None = teacher forcing (TrainingHelper).  {'prog': [B,L] ids, 'act': [B,k,T] ids} = the
    decoder INPUT ids actually fed at each step under scheduled sampling
    (ScheduledEmbeddingTrainingHelper, model_full.py:414-423): a mixture of ground-truth tokens
    and draws from the decoder's own predictions.  The draws are random (and not differentiated
    through), so for a parity check they are taken from the implementation under test and the
    oracle reproduces everything that follows from them.

    p: name -> tensor (see param_shapes).  batch: dict with the reference's
    batch_chunk keys (model_full.py:185-206) as torch tensors:
      s_h [B,k,T,H,W,C] float, a_h [B,k,T,A] float one-hot, a_h_tokens [B,k,T] int,
      per [B,k,T,P] float, program [B,V,L] float one-hot, program_tokens [B,L] int,
      program_len [B,1], demo_len [B,k].
    Returns a dict with loss terms, logits and the per-call BN batch statistics."""
    dt = p['conv1/W'].dtype
    B, k, T, L = cfg.batch_size, cfg.k, cfg.max_demo_len, cfg.max_program_len
    U = cfg.num_lstm_cell_units
    s_h = batch['s_h'].to(dt)
    demo_len = batch['demo_len'].to(torch.int64)             # cast, model_full.py:165
    program_len = batch['program_len'].to(torch.int64).reshape(B)
    bn_stats = {}

    def state_encoder(s, tag):                               # model_full.py:216-231
        x = s
        for l in range(1, cfg.n_conv + 1):
            x, m, v = conv2d_lrelu_bn(x, p['conv%d/W' % l], p['conv%d/b' % l],
                                      p['conv%d/beta' % l], p['conv%d/gamma' % l],
                                      moving=None if moving is None else moving['conv%d' % l])
            bn_stats.setdefault('conv%d' % l, []).append((m, v))
        return x.reshape(x.shape[0], -1)

    # ---- Demo_Encoder, called k times (model_full.py:235-258,373-379)
    step1_hist, step1_h, step1_c = [], [], []
    for i in range(k):
        frames = s_h[:, i].reshape(B * T, cfg.h, cfg.w, cfg.depth)
        feats = state_encoder(frames, i).reshape(B, T, -1)
        outs, h, c = dynamic_rnn(feats, demo_len[:, i],
                                 p['demo_lstm/kernel'], p['demo_lstm/bias'])
        step1_hist.append(outs)
        step1_h.append(h)
        step1_c.append(c)
    model = getattr(cfg, 'model', 'full')
    if model not in ('full', 'summarizer', 'synthesis_baseline'):
        raise ValueError(model)
    if model == 'synthesis_baseline':
        # models/baselines/model_synthesis.py:324-358: one encoder pass, demonstrations pooled
        return _baseline_tail(p, batch, cfg, torch.stack(step1_h, dim=1), torch.stack(step1_c, dim=1),
                              None, None, bn_stats, fed_ids, moving)
    summary_h = torch.stack(step1_h, dim=1).mean(dim=1)      # :380-385 avgpool
    summary_c = torch.stack(step1_c, dim=1).mean(dim=1)

    # ---- SecondPathEncoder, called k times (model_full.py:260-277,387-398)
    demo_h, demo_c = [], []
    for i in range(k):
        _, h, c = dynamic_rnn(step1_hist[i], demo_len[:, i],
                              p['second_lstm/kernel'], p['second_lstm/bias'],
                              c0=summary_c, h0=summary_h)
        demo_h.append(h)
        demo_c.append(c)
    stack_h = torch.stack(demo_h, dim=1)
    stack_c = torch.stack(demo_c, dim=1)
    if model == 'summarizer':
        # models/baselines/model_summarizer.py:345-352,389-394: relation network ONLY (no avg term)
        return _baseline_tail(p, batch, cfg, stack_h, stack_c, summary_h, summary_c, bn_stats, fed_ids, moving)
    demo_h_summary = stack_h.mean(dim=1) + rn_pool(stack_h, p, 'rn_h', moving)   # :399-404
    demo_c_summary = stack_c.mean(dim=1) + rn_pool(stack_c, p, 'rn_c', moving)

    def shift_tokens(tokens, token_dim):                     # model_full.py:447-450
        s_tok = torch.full((tokens.shape[0], 1), token_dim + 1, dtype=tokens.dtype)
        return torch.cat([s_tok, tokens[:, :-1]], dim=1)

    # ---- Program decoder (model_full.py:497-511)
    V = cfg.dim_program_token
    ptoks = shift_tokens(batch['program_tokens'].to(torch.int64), V)
    if fed_ids is not None:
        ptoks = fed_ids['prog'].to(torch.int64)
    pemb = embedding_lookup_oob0(p['prog/embedding'], ptoks)
    pred_program = training_decoder(pemb, program_len, demo_c_summary, demo_h_summary,
                                    p['prog/lstm/kernel'], p['prog/lstm/bias'],
                                    p['prog/proj'], L)

    # ---- Action decoders, k calls (model_full.py:525-545)
    A = cfg.action_space
    pred_action = []
    for i in range(k):
        atoks = shift_tokens(batch['a_h_tokens'][:, i].to(torch.int64), A)
        if fed_ids is not None:
            atoks = fed_ids['act'][:, i].to(torch.int64)
        aemb = embedding_lookup_oob0(p['act/embedding'], atoks)
        pred_action.append(training_decoder(
            aemb, demo_len[:, i], demo_c[i], demo_h[i],
            p['act/lstm/kernel'], p['act/lstm/bias'], p['act/proj'], T))

    # ---- Perception decoders, k calls (model_full.py:564-583; inputs NOT shifted,
    #      Per_Encoder = fc + BN without activation, :308-316)
    P = cfg.per_dim
    pred_per = []
    per = batch['per'].to(dt)
    for i in range(k):
        pin, m, v = fc_lrelu_bn(per[:, i], p['per/fc/W'], p['per/fc/b'],
                                p['per/fc/beta'], p['per/fc/gamma'], act=False,
                                moving=None if moving is None else moving['per/fc'])
        bn_stats.setdefault('per/fc', []).append((m, v))
        pred_per.append(training_decoder(
            pin, demo_len[:, i], demo_c[i], demo_h[i],
            p['per/lstm/kernel'], p['per/lstm/bias'], p['per/proj'], T))

    # ---- Losses (model_full.py:921-932,1014-1038,1061-1079)
    program_loss = sequence_loss(pred_program, batch['program'].to(dt), program_len,
                                 L, V, 'program')
    gt_act = batch['a_h'].to(dt).permute(0, 1, 3, 2)         # :323-326 -> [B,k,A,T]
    gt_per = per.permute(1, 0, 3, 2)                         # :331     -> [k,B,P,T]
    action_losses = [sequence_loss(pred_action[i], gt_act[:, i], demo_len[:, i], T, A, 'action')
                     for i in range(k)]
    per_losses = [sequence_loss(pred_per[i], gt_per[i], demo_len[:, i], T, P, 'per')
                  for i in range(k)]
    avg_action_loss = sum(action_losses) / k
    avg_per_loss = sum(per_losses) / k
    loss = program_loss + avg_action_loss + avg_per_loss
    return dict(loss=loss, program_loss=program_loss, avg_action_loss=avg_action_loss,
                avg_per_loss=avg_per_loss, pred_program=pred_program,
                pred_action=torch.stack(pred_action, dim=1),   # [B,k,A,T]
                pred_per=torch.stack(pred_per, dim=1),         # [B,k,P,T]
                summary_h=summary_h, summary_c=summary_c,
                demo_h=stack_h, demo_c=stack_c,
                demo_h_summary=demo_h_summary, demo_c_summary=demo_c_summary,
                bn_stats=bn_stats)


def _baseline_tail(p, batch, cfg, stack_h, stack_c, summary_h, summary_c, bn_stats, fed_ids, moving):
    """Program decoder + program loss of the two program-synthesis baselines
    (model_summarizer.py:500-518,834-847; model_synthesis.py:461-479,795-808): the only loss term."""
    dt = stack_h.dtype
    B, L, V = cfg.batch_size, cfg.max_program_len, cfg.dim_program_token
    program_len = batch['program_len'].to(torch.int64).reshape(B)
    if cfg.model == 'summarizer':
        demo_h_summary = rn_pool(stack_h, p, 'rn_h', moving)
        demo_c_summary = rn_pool(stack_c, p, 'rn_c', moving)
    elif cfg.demo_aggregation == 'avgpool':
        demo_h_summary, demo_c_summary = stack_h.mean(dim=1), stack_c.mean(dim=1)
    elif cfg.demo_aggregation == 'maxpool':
        demo_h_summary, demo_c_summary = stack_h.max(dim=1).values, stack_c.max(dim=1).values
    else:
        # 'concat' hands a [B, k*U] state to a U-unit cell (model_synthesis.py:339-341,463-467)
        raise ValueError('Unknown demo aggregation type')
    tokens = batch['program_tokens'].to(torch.int64)
    ptoks = torch.cat([torch.full((B, 1), V + 1, dtype=tokens.dtype), tokens[:, :-1]], dim=1)
    if fed_ids is not None:
        ptoks = fed_ids['prog'].to(torch.int64)
    pemb = embedding_lookup_oob0(p['prog/embedding'], ptoks)
    pred_program = training_decoder(pemb, program_len, demo_c_summary, demo_h_summary,
                                    p['prog/lstm/kernel'], p['prog/lstm/bias'], p['prog/proj'], L)
    program_loss = sequence_loss(pred_program, batch['program'].to(dt), program_len, L, V, 'program')
    zero = program_loss.detach() * 0
    return dict(loss=program_loss, program_loss=program_loss, avg_action_loss=zero, avg_per_loss=zero,
                pred_program=pred_program, summary_h=summary_h, summary_c=summary_c,
                demo_h=stack_h, demo_c=stack_c, demo_h_summary=demo_h_summary,
                demo_c_summary=demo_c_summary, bn_stats=bn_stats)


def loss_and_grads(params, batch, cfg, dtype=torch.float32, fed_ids=None):
    """Forward + torch autograd.  Returns (outputs dict (detached), grads dict)."""
    p = {n: torch.as_tensor(v).detach().clone().to(dtype).requires_grad_(True)
         for n, v in params.items()}
    out = forward(p, batch, cfg, fed_ids=fed_ids)
    out['loss'].backward()
    grads = {n: (t.grad.detach() if t.grad is not None else torch.zeros_like(t))
             for n, t in p.items()}
    res = {}
    for n, v in out.items():
        res[n] = v.detach() if torch.is_tensor(v) else v
    return res, grads


# --------------------------------------------------------------------------- optimizer

def exponential_decay_staircase(lr, step, decay_steps=10000, decay_rate=0.5):
    """[TF-1.3] tf.train.exponential_decay(staircase=True) (trainer.py:84-91; D13)."""
    return lr * decay_rate ** (step // decay_steps)


def polynomial_decay(start, step, decay_steps, end, power=1.0):
    """[TF-1.3] tf.train.polynomial_decay, cycle=False (model_full.py:64-67; D13)."""
    s = min(step, decay_steps)
    return (start - end) * (1 - s / decay_steps) ** power + end


def adam_clip_step(params, grads, m, v, step, lr, clip=20.0, b1=0.9, b2=0.999, eps=1e-8):
    """[TF-1.3] optimize_loss(clip_gradients=20.0, AdamOptimizer) (trainer.py:102-109;
    SURVEY D12): clip_by_global_norm then Adam with
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps).  ``step`` is the
    1-based Adam timestep.  Updates dicts in place; returns the global norm."""
    sq = sum(float((g.double() ** 2).sum()) for g in grads.values())
    norm = math.sqrt(sq)
    scale = clip / max(norm, clip)
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    for n in params:
        g = grads[n] * scale
        m[n] = b1 * m[n] + (1 - b1) * g
        v[n] = b2 * v[n] + (1 - b2) * g * g
        params[n] = params[n] - lr_t * m[n] / (v[n].sqrt() + eps)
    return norm


# --------------------------------------------------------------------------- greedy decoding (N1)

def greedy_decoder(table, c0, h0, kernel, bias, proj, start_id, end_id, max_len):
    """[TF-1.3] BasicDecoder(cell, GreedyEmbeddingHelper(embedding, start_tokens, end_token),
    (c0, h0), Dense(no bias)) under dynamic_decode(impute_finished=False,
    maximum_iterations=max_len)  (models/model_full.py:424-435,465-490).

    start_tokens = token_dim (the last, in-range embedding row, :426-427).  Each step:
    logits = Dense(cell(inputs)); sample = argmax(logits) (first index on ties);
    finished |= (sample == end_token); the next input is embedding(sample) for EVERY row
    (finished rows keep running on their own samples until all rows are finished or max_len is
    reached); sequence_length[r] = step+1 at which r first finished (max_len if never).
    Returns (logits [B, token_dim, max_len] zero-padded past the steps that ran,
             sample ids [B, max_len] (zeros past the steps that ran), lengths [B])."""
    B = c0.shape[0]
    token_dim = proj.shape[1]
    c, h = c0, h0
    inputs = table[torch.full((B,), start_id, dtype=torch.int64)]
    finished = torch.zeros(B, dtype=torch.bool)
    lengths = torch.zeros(B, dtype=torch.int64)
    logits_l, ids_l = [], []
    for t in range(max_len):
        if bool(finished.all()):
            break
        c, h = basic_lstm_cell(inputs, c, h, kernel, bias)
        logits = h @ proj
        sample = torch.argmax(logits, dim=-1)
        logits_l.append(logits)
        ids_l.append(sample)
        step_finished = sample == end_id
        next_finished = finished | step_finished | (t + 1 >= max_len)
        lengths = torch.where(~finished & next_finished, torch.full_like(lengths, t + 1), lengths)
        finished = next_finished
        inputs = table[sample]
    n = len(logits_l)
    out = torch.zeros(B, max_len, token_dim, dtype=c0.dtype)
    ids = torch.zeros(B, max_len, dtype=torch.int64)
    if n:
        out[:, :n] = torch.stack(logits_l, dim=1)
        ids[:, :n] = torch.stack(ids_l, dim=1)
    return out.permute(0, 2, 1), ids, lengths


def sequence_stats(pred, gt, pred_len, gt_len, max_len, token_dim):
    """The accuracy statistics of Sequence_Loss (models/model_full.py:626-683):
    token_acc = sum(eq * min_mask) / sum(max_mask); is_same_seq = all positions equal under the
    gt mask AND equal lengths; seq_acc = mean(is_same_seq).  pred, gt: [B, token_dim, max_len]."""
    B = pred.shape[0]
    pos = torch.arange(max_len).unsqueeze(0)
    gt_mask = (pos < gt_len.unsqueeze(1)).to(pred.dtype)
    max_mask = (pos < torch.maximum(pred_len, gt_len).unsqueeze(1)).to(pred.dtype)
    min_mask = (pos < torch.minimum(pred_len, gt_len).unsqueeze(1)).to(pred.dtype)
    label_argmax = gt.permute(0, 2, 1).argmax(dim=-1)
    logit_argmax = pred.permute(0, 2, 1).argmax(dim=-1)
    eq = (label_argmax == logit_argmax).to(pred.dtype)
    token_acc = (eq * min_mask).sum() / max_mask.sum()
    seq_equal = (label_argmax.to(pred.dtype) * gt_mask) == (logit_argmax.to(pred.dtype) * gt_mask)
    is_same_seq = seq_equal.all(dim=-1) & (gt_len == pred_len)
    return dict(token_acc=token_acc, seq_acc=is_same_seq.to(pred.dtype).mean(),
                is_same_seq=is_same_seq, pred_tokens=logit_argmax)


def greedy_program_and_actions(p, batch, cfg, fwd):
    """Greedy twins of the program and action decoders (models/model_full.py:513-523,546-558)
    given a finished `forward` result (for the decoder initial states)."""
    B, k, T, L = cfg.batch_size, cfg.k, cfg.max_demo_len, cfg.max_program_len
    V, A = cfg.dim_program_token, cfg.action_space
    gp, gp_ids, gp_len = greedy_decoder(p['prog/embedding'], fwd['demo_c_summary'], fwd['demo_h_summary'],
                                        p['prog/lstm/kernel'], p['prog/lstm/bias'], p['prog/proj'],
                                        start_id=V, end_id=3, max_len=L)     # 'm)' == 3
    if getattr(cfg, 'model', 'full') != 'full':                               # baselines: program only
        return dict(greedy_pred_program=gp, greedy_program_ids=gp_ids, greedy_pred_program_len=gp_len)
    acts = [greedy_decoder(p['act/embedding'], fwd['demo_c'][:, i], fwd['demo_h'][:, i],
                           p['act/lstm/kernel'], p['act/lstm/bias'], p['act/proj'],
                           start_id=A, end_id=A - 1, max_len=T) for i in range(k)]
    return dict(greedy_pred_program=gp, greedy_program_ids=gp_ids, greedy_pred_program_len=gp_len,
                greedy_pred_action=torch.stack([a[0] for a in acts], dim=1),      # [B,k,A,T]
                greedy_action_ids=torch.stack([a[1] for a in acts], dim=1),       # [B,k,T]
                greedy_pred_action_len=torch.stack([a[2] for a in acts], dim=1))  # [B,k]
