import os, sys, torch
sys.path.insert(0, '/root/repo')
def run(flag):
    os.environ['D2P_TOKEN_PROJECTION'] = flag
    from demo2program_amd.config import make_config
    from demo2program_amd.trainer import Trainer
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel')
    tr = Trainer(cfg, make_train_dir=False)
    feeds = [tr.model.get_feed_dict(make_batch(cfg, seed=123 + i)) for i in range(4)]
    out = []
    for i in range(12):
        out.append(float(tr.train_step(feeds[i % 4]).item()))
    g = tr.model.params.grad.clone()
    return out, g, tr
a, ga, ta = run('1')
b, gb, tb = run('0')
for i, (x, y) in enumerate(zip(a, b)):
    print('step %2d  token-proj %.7f  gathered %.7f  diff %.2e' % (i, x, y, abs(x - y)))
# gradient comparison on identical parameters: one step from the same init
import numpy as np
def one_step_grads(flag):
    os.environ['D2P_TOKEN_PROJECTION'] = flag
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel')
    m = Model(cfg, seed=123)
    feed = m.get_feed_dict(make_batch(cfg, seed=123))
    m.forward(feed); m.backward()
    return {n: m.params.g[n].clone() for n in m.params.g}
g1, g0 = one_step_grads('1'), one_step_grads('0')
worst = []
for n in g1:
    sc = g0[n].abs().max().item() + 1e-30
    worst.append(((g1[n] - g0[n]).abs().max().item() / sc, n))
worst.sort(reverse=True)
print('largest relative gradient differences (max|d| / max|g|):', worst[:6])
