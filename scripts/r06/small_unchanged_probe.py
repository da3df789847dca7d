"""The UNCHANGED caller (examples/train_deblur_synth.py DeblurTrainer.iteration_unchanged: train.py:430-807 as written, after the
import swaps alone) at the reference's own operating point -- 512x288, 20 k + 10 k splats: ms per iteration, and where the host's
time goes (cProfile, top entries by own time)."""
import cProfile, gc, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_deblur_synth as TD
torch.autograd.set_multithreading_enabled(False)
for lam in (0.0, 1e-2):
    tr = TD.DeblurTrainer("cuda:0", 20_000, 10_000, 512, 288, 2, iters=10000, lambda_flow=lam)
    tr.iteration_unchanged()
    gc.collect(); gc.freeze()
    for _ in range(4):
        tr.iteration_unchanged()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        tr.iteration_unchanged()
    torch.cuda.synchronize()
    print(f"512x288 unchanged caller, lambda_flow {lam}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per iteration")
    if lam == 0.0 and "--profile" in sys.argv:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(10):
            tr.iteration_unchanged()
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    del tr
