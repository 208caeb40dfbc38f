import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fractal_renderer_amd as fr
fr.init(0)
cfg = fr.Config.new(); cfg.width = cfg.height = 16384; cfg.iterations = 1024; cfg.pos.re = -0.6; cfg.exposure = 5.0
buf = np.empty((16384, 16384, 3), dtype=np.uint8)
for _ in range(3):
    t0 = time.perf_counter(); fr.get_image_rows(cfg, 0, 16384, out=buf); print("resident %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
