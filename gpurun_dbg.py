import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("QN", "4"))
import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sedef_amd, bench
n = 100000
pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 1000, seed=42)
words, q_word, t_word = bench.pack_batch(pool, q_off, qlen, t_off, tlen)
tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
tasks["w"], tasks["zdrop"] = 128, -1
dev = torch.device("cuda", 0)
d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
cells = 24046888294
want = 3
def mk():
    eng = sedef_amd.Extz2Engine(0, 48 << 30)
    d_out = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_cig = torch.empty(256 * n, dtype=torch.int32, device=dev)
    return eng, d_out, d_cig
lanes = [mk() for _ in range(int(sys.argv[1]))]
def run(lane, k):
    eng, d_out, d_cig = lane
    for _ in range(k):
        eng.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), 256 * n, want=want)
for lane in lanes: run(lane, 1)
torch.cuda.synchronize()
K = 12
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(lane, K // len(lanes))) for lane in lanes]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("lanes", len(lanes), "steps", K, "ms/step %.3f" % (dt / K * 1e3), "Gcell/s %.1f" % (cells * K / dt / 1e9))
