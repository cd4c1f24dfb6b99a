"""Round 6: does anything read device memory it never wrote?  Fills most of the free device memory with a pattern (NaN / large
values), frees it, then runs the given GPU tests in THIS process, again and again.  A test that passes on fresh (zeroed) memory and
fails here reads uninitialised memory.     python scripts/r6_poison_probe.py [rounds] [-k expression]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pytest
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
expr = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "-k" else "overlap_pipeline_survives or contiguous_split_on_device"
rc_all = 0
for r in range(rounds):
    free, _ = torch.cuda.mem_get_info(0)
    n = int(free * 0.85) // 4
    t = torch.empty(n, dtype=torch.float32, device="cuda:0")
    t.fill_(float("nan") if r % 2 == 0 else 3.0e38)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
    rc = pytest.main([os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k", expr, "-p", "no:cacheprovider"])
    print("round", r, "rc", rc, flush=True)
    rc_all |= int(rc)
sys.exit(rc_all)
