"""Which allocation makes the caching allocator go to hipMalloc in the first iterations of the trainer?
Records the allocator's history over iterations 1..4 and prints every segment_alloc (= device allocation) with the Python
frames that asked for it."""
import gc, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import train_deblur_synth as TD
torch.autograd.set_multithreading_enabled(False)
tr = TD.DeblurTrainer("cuda:0", 200_000, 100_000, 1352, 1014, 2, iters=10000)
tr.iteration(); torch.cuda.synchronize()
gc.collect(); gc.freeze()
torch.cuda.memory._record_memory_history(max_entries=400000, context="alloc", stacks="python")
marks = []
for i in range(1, 6):
    tr.iteration(); torch.cuda.synchronize()
    ms = torch.cuda.memory_stats()
    marks.append((i, ms["num_device_alloc"], ms["reserved_bytes.all.current"] >> 20))
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
print(marks)
for tr_ in snap["device_traces"]:
    for e in tr_:
        if e["action"] == "segment_alloc":
            fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in e.get("frames", []) if "mobgs" in f["filename"] or "train_deblur" in f["filename"]]
            print("segment_alloc", e["size"] >> 20, "MB", " <- ".join(fr[:6]))
# requested sizes of the biggest allocations per iteration
big = {}
for tr_ in snap["device_traces"]:
    for e in tr_:
        if e["action"] == "alloc" and e["size"] > (256 << 20):
            fr = [f"{os.path.basename(f['filename'])}:{f['line']}" for f in e.get("frames", []) if "mobgs" in f["filename"]]
            big.setdefault(" <- ".join(fr[:3]), []).append(e["size"] >> 20)
for k, v in big.items():
    print(len(v), "allocs > 256 MB at", k, "sizes MB:", sorted(set(v)))
