"""Per-workgroup run times of the value-net kernel (diagnostic build: tools/ab_build.sh wgclock "-DBG_EVAL_WGCLOCK", BGAMD_LIB=.../libbgamd_wgclock.so):
every workgroup stores, at the top end of the value array, the microseconds between its first and last instruction and the tiles its waves drew.
Prints their distribution for a few steady-state steps at 65 536 lanes -> how far the slowest workgroup runs behind the mean (the launch ends with it)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, "backgammon-engine_amd")
import backgammon_env as bg
from backgammon_env import _capi
w = np.fromfile("tests/golden/tdgammonNEW100k.f32", dtype=np.float32)
n = 65536
cap = n * 512
env = bg.VecGame(n, device=0, seed=20240603, arena_rows=cap)
env.load_weights(w)
env.run_greedy(180)
G = 256
for rep in range(5):
    env.run_greedy(3)
    val = torch.empty((2 * G,), dtype=torch.float32, device="cuda")
    _capi.check(env._lib.bgamd_env_unique_rows_read(env._h, cap - 2 * G, 2 * G, None, C.c_void_p(val.data_ptr()), None), "read")
    torch.cuda.synchronize()
    v = val.cpu().numpy()
    us, tiles = v[G:][::-1], v[:G][::-1]
    xv = torch.empty((3072,), dtype=torch.float32, device="cuda")
    _capi.check(env._lib.bgamd_env_unique_rows_read(env._h, cap - 3072, 3072, None, C.c_void_p(xv.data_ptr()), None), "read")
    torch.cuda.synchronize()
    x = xv.cpu().numpy()[::-1]                          # x[k] = values[cap - 1 - k]
    xt, xr = x[1024:1536], x[2048:2560]                 # expand_all_kernel: run time and rows of workgroup 0 .. 511 (the first 256: doubles turns)
    ND = 316                                                # doubles workgroups of a 65 536-lane launch (62 % of 512, a multiple of the four list parts)
    for name, sl in (("doubles workgroups", slice(0, ND)), ("non-doubles workgroups", slice(ND, 512))):
        t, r = xt[sl], xr[sl]
        print("   expand_all %s: run time us min %.1f mean %.1f max %.1f sigma %.2f; rows per workgroup min %d mean %.0f max %d; corr(time, rows) %.2f"
              % (name, t.min(), t.mean(), t.max(), t.std(), r.min(), r.mean(), r.max(), np.corrcoef(t, r)[0, 1]), flush=True)
    pv = torch.empty((4096,), dtype=torch.float32, device="cuda")
    _capi.check(env._lib.bgamd_env_unique_rows_read(env._h, cap - 4096 - 4096, 4096, None, C.c_void_p(pv.data_ptr()), None), "read")
    torch.cuda.synchronize()
    pp = pv.cpu().numpy()[::-1].reshape(512, 8)[:, :4]          # pp[b][k] = values[cap - 1 - 4096 - 8 b - k]: node logic, scan, allocation, successors
    for name, sl in (("doubles", slice(0, ND)), ("non-doubles", slice(ND, 512))):
        m = pp[sl].mean(0)
        print("   expand_all %s workgroups, thread 0's time per stage (us, summed over the phases): node logic %.1f  scan + table %.1f  allocation %.1f  successors %.1f  (sum %.1f)"
              % (name, m[0], m[1], m[2], m[3], m.sum()), flush=True)
    print("step %d: workgroup run time us: min %.1f mean %.1f max %.1f (max - mean %.1f, sigma %.2f); tiles drawn per workgroup: min %d mean %.1f max %d; corr(time, tiles) %.2f"
          % (rep, us.min(), us.mean(), us.max(), us.max() - us.mean(), us.std(), tiles.min(), tiles.mean(), tiles.max(), np.corrcoef(us, tiles)[0, 1]), flush=True)
