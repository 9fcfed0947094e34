"""The multi-GPU exchange step behind the C ABI (include/sedef_hip.h: sdf_comm_*, sdf_allgatherv_results) and the C++ driver
above it (sedef_amd/bin/sdf_multi: one batch sharded over the GPUs of a node, RCCL all-gatherv of the results) -- as far as a
box with ONE GPU can execute them: a communicator of one rank, the count all-gather on RCCL, and (SDF_COMM_SELFTEST=1) the
rank's own ranges through ncclSend / ncclRecv to itself."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("selftest", ["0", "1"])
def test_allgatherv_results_one_rank(monkeypatch, selftest):
    import torch

    import bench
    import sedef_amd
    monkeypatch.setenv("SDF_COMM_SELFTEST", selftest)
    lib = sedef_amd.load_library()
    lib.sdf_comm_create.restype = C.c_void_p
    lib.sdf_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.sdf_comm_last_error.restype = C.c_char_p
    lib.sdf_comm_last_error.argtypes = [C.c_void_p]
    lib.sdf_comm_destroy.argtypes = [C.c_void_p]
    lib.sdf_allgatherv_results.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    uid = (C.c_ubyte * 128)()
    assert lib.sdf_comm_unique_id(uid, 128) == 0, lib.sdf_comm_last_error(None)
    comm = lib.sdf_comm_create(0, 1, 0, uid)
    assert comm, lib.sdf_comm_last_error(None)
    try:
        n = 3000
        pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 300, seed=5)
        words, q_word, t_word = bench.pack_batch(pool, q_off, qlen, t_off, tlen)
        tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
        tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"], tasks["w"], tasks["zdrop"] = q_word, t_word, qlen, tlen, 64, -1
        dev = torch.device("cuda", 0)
        eng = sedef_amd.Extz2Engine(0)
        d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
        cap = 128 * n
        d_out = torch.zeros(n * 16, dtype=torch.int32, device=dev)
        d_cig = torch.zeros(cap, dtype=torch.int32, device=dev)
        used = eng.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cap,
                                      want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
        all_out = torch.full((n * 16,), -1, dtype=torch.int32, device=dev)
        all_cig = torch.full((cap,), -1, dtype=torch.int32, device=dev)
        counts = np.zeros(2, np.uint64)
        torch.cuda.synchronize()
        # a buffer that is too small: the sizes come back
        rc = lib.sdf_allgatherv_results(comm, d_out.data_ptr(), n, d_cig.data_ptr(), used, all_out.data_ptr(), n - 1,
                                        all_cig.data_ptr(), cap, counts.ctypes.data, None)
        assert rc == -5 and counts.tolist() == [n, used]
        rc = lib.sdf_allgatherv_results(comm, d_out.data_ptr(), n, d_cig.data_ptr(), used, all_out.data_ptr(), n,
                                        all_cig.data_ptr(), cap, counts.ctypes.data, None)
        assert rc == 0, lib.sdf_comm_last_error(comm)
        torch.cuda.synchronize()
        assert counts.tolist() == [n, used] and used > n
        assert torch.equal(all_out, d_out) and torch.equal(all_cig[:used], d_cig[:used])
    finally:
        lib.sdf_comm_destroy(comm)


@pytest.mark.parametrize("selftest", ["0", "1"])
def test_cpp_driver_shards_aligns_and_gathers(selftest):
    """sedef_amd/bin/sdf_multi on the devices of this box (one): shard by cells, sdf_extz2_batch_device per device thread,
    sdf_allgatherv_results, union check; exit code 0 and one JSON line."""
    exe = os.path.join(ROOT, "sedef_amd", "bin", "sdf_multi")
    env = dict(os.environ, SDF_COMM_SELFTEST=selftest, GPU_MAX_HW_QUEUES="8")
    p = subprocess.run([exe, "--devices", "0", "--tasks", "20000", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["tasks"] == 20000 and d["value"] > 50
    assert "FAILED" not in d["union_check"] and abs(d["shard_balance_max_over_mean"] - 1.0) < 1e-6
