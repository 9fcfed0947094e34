import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/profiles")
import bench, sedef_amd, shapes_bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
w = int(sys.argv[3]) if len(sys.argv) > 3 else 512
eng = sedef_amd.Extz2Engine(0, 16 << 30)
shapes_bench.run("%d x %d^2 w=%d" % (n, L, w), bench.synth_batch(n, L, 7), w, eng, torch.device("cuda", 0))
