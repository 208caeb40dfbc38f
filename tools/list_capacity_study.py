import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.argv = ["c4_ab.py", "11"]
import ctypes as C
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import fractal_renderer_amd as fr
from fractal_renderer_amd import _native
lib = _native.load()
for cap in (0, 1 << 20, 1 << 19, 1 << 18, 0, 1 << 20):
    lib.fr_debug_set_two_pass_capacity(cap)
    print("== list capacity", cap, flush=True)
    exec(open(os.path.join(ROOT, "tools", "c4_ab.py")).read())
