import sys, time, torch
sys.path.insert(0, '/root/repo')
from demo2program_amd.config import make_config
from demo2program_amd.trainer import Trainer
from demo2program_amd.synthetic import make_batch
from demo2program_amd.lib import load
lib = load()
cfg = make_config('karel')
for fw, bw, pipe in ((256, 256, 1), (512, 256, 1), (256, 512, 1), (512, 512, 1), (384, 384, 1), (256, 256, 0), (192, 192, 1)):
    lib.d2p_lstm_set_tiling(fw, bw, pipe)
    tr = Trainer(cfg, make_train_dir=False)
    feed = tr.model.get_feed_dict(make_batch(cfg, seed=1))
    for _ in range(8): tr.train_step(feed)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(40): tr.train_step(feed)
    torch.cuda.synchronize(); dt = (time.time() - t) / 40
    print(fw, bw, pipe, '%.3f ms' % (dt * 1e3), flush=True)
    del tr
