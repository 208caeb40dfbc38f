import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import fractal_renderer_amd as fr
fr.init(0)
def call(w, h, it, tag, ch=3, tile=None):
    cfg = fr.Config.new()
    cfg.width, cfg.height, cfg.iterations = w, h, it
    buf = np.zeros((h, w, ch), dtype=np.uint8); buf.fill(1)
    t0 = time.perf_counter()
    if ch == 3:
        fr.get_image_rows(cfg, 0, h, fr.Precision.F64, out=buf)
    else:
        fr.get_image_rgba(cfg, fr.Precision.F64, out=buf)
    print("%s %dx%d: %.3f ms" % (tag, w, h, (time.perf_counter() - t0) * 1e3), flush=True)
call(750, 500, 50, "warm small")
call(1920, 1080, 1024, "1080p")
call(2600, 1500, 1024, "11.7 MB")
call(2800, 1700, 1024, "14.3 MB")
call(3000, 1900, 1024, "17.1 MB")
call(3840, 2160, 1024, "4K first")
call(3840, 2160, 1024, "4K second (new buffer)")
call(5000, 3000, 1024, "45 MB")
call(5000, 3000, 1024, "45 MB again")
