import random, sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
import torch
import parity_util, seqgen, sharded_util, repair_util, autocycler_amd
from autocycler_amd import sharded
dev = torch.device("cuda", 0)
t0 = time.time(); n = 0; fails = 0
for seed in range(24, 424):
    k = random.Random(seed).choice([3, 5, 7, 9, 11, 13, 21, 31, 51, 51, 51, 77, 101, 123, 151, 251])
    seqs, fn, hd = seqgen.make_case(seed, k)
    try:
        parity_util.check_case(k, seqs, fn, hd, repair=(seed % 3 != 0), distances=(seed % 2 == 0))
        if seed % 4 == 0:
            sharded_util.run_case(autocycler_amd.LIB_PATH, k, seqs, fn, hd, sharded.Comm(dev), dev)
        if seed % 5 == 0:
            repair_util.check_repair(autocycler_amd.LIB_PATH, k, seqs, fn, hd, device="cuda:0")
    except Exception as e:
        fails += 1; print("FAIL seed", seed, "k", k, repr(e)[:300])
        if fails > 5: break
    n += 1
print("gpu fuzz cases", n, "fails", fails, "time %.1f" % (time.time() - t0))
