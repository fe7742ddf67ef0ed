"""Eager vs HIP-graph replay of the greedy step (the step is capturable through torch.cuda.graph)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "backgammon-engine_amd")]
import backgammon_env as bg
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
for n in (512, 4096, 65536):
    env = bg.VecGame(n, seed=3, arena_rows=max(n * 512, 1 << 20)); env.load_weights(w)
    for _ in range(200): env.step_greedy()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): env.step_greedy()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 400
    side, g = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(8): env.step_greedy()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 400
    print(f"n={n}: eager {eager*1e6:.1f} us/step ({n/eager/1e6:.1f} M steps/s), graph (8 steps per replay) {graph*1e6:.1f} us/step ({n/graph/1e6:.1f} M steps/s)", flush=True)
