"""Full-size (1352x1014, 300 k) training iterations over a longer stretch, eager then graphed with the example loop's polling: time per
iteration in windows, allocator state, recaptures."""
import gc, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_deblur_synth as TD
from mobgs_amd.graphed import GraphedCallable
torch.autograd.set_multithreading_enabled(False)


def mem():
    s = torch.cuda.memory_stats()
    return f"allocated {torch.cuda.memory_allocated() >> 20} MB, reserved {torch.cuda.memory_reserved() >> 20} MB, device mallocs {s.get('num_device_alloc', 0)}"


for graph in (False, True):
    tr = TD.DeblurTrainer("cuda:0", 200_000, 100_000, 1352, 1014, 2, iters=10000, lambda_flow=1e-2)
    tr.iteration(); tr.iteration()
    gc.collect(); gc.freeze()
    fb = GraphedCallable(tr.forward_backward, warmup=0) if graph else None
    torch.cuda.synchronize()
    print(f"1352x1014 / 300 k training loop ({'graphed' if graph else 'eager'}):", mem(), flush=True)
    pending, rec = [], 0
    for w in range(4):
        t0 = time.perf_counter()
        for _ in range(25):
            if fb is not None:
                loss = fb(); tr.optimizer_step()
                ev = torch.cuda.Event(); ev.record(); pending.append(ev)
                if len(pending) > 2:
                    pending.pop(0).synchronize()
                if not fb.check():
                    torch.cuda.synchronize(); fb.recapture(); rec += 1
            else:
                loss = tr.iteration()
        torch.cuda.synchronize()
        print(f"  iterations {25 * w:3d}..{25 * w + 24}: {(time.perf_counter() - t0) / 25 * 1e3:.2f} ms per iteration, loss {float(loss):.5f}, recaptures {rec}", flush=True)
    print("  end:", mem(), flush=True)
    del tr, fb
    gc.collect(); torch.cuda.empty_cache()
