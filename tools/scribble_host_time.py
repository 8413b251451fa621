import sys, time, random
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import vpu_oracle as vo
from pvpuformer_amd.isegm.engine import prompt_sim as ps
from pvpuformer_amd.isegm.model.scribble import scribble_profiles, scribble_curves
b = vo.synth_batch(12, 448, seed=3)
gt = b["instances"][:, 0].numpy() > 0.5
rng, nr = random.Random(0), np.random.RandomState(0)
scr, rects = ps.cal_scribble(gt, rng=rng, np_rng=nr)
t = time.time()
for _ in range(10): scr, rects = ps.cal_scribble(gt, rng=rng, np_rng=nr)
t1 = (time.time() - t) / 10
t = time.time()
for _ in range(10): scribble_profiles(scr, rects, 448, rng); scribble_curves(scr)
print("box host: cal_scribble %.1f ms, profiles+curves %.1f ms" % (t1 * 1e3, (time.time() - t) / 10 * 1e3))
