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
print("gpu fuzz cases", n, "fails", fails, "time %.1f" % (time.time() - t0), flush=True)
# random synthetic assembly sets (the CPU sweep's second part, on the device): long novel runs, multi-wavefront unitigs, real
# expand_repeats levels — what the small adversarial cases do not reach
from autocycler_amd import synth
t0 = time.time(); m = 0
for seed in range(1000, 1400):
    r = random.Random(seed)
    k = r.choice([21, 31, 51, 51, 77, 101])
    na = r.randint(2, 10); genome = r.choice([8000, 20000, 50000, 120000]); plasmid = r.choice([0, 1500, 4000])
    sub = r.choice([1e-4, 1e-3, 5e-3, 2e-2]); indel = r.choice([0, 1e-5, 1e-4, 1e-3])
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(na, genome=genome, plasmid=plasmid, sub=sub, indel=indel, seed=seed)):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    try:
        parity_util.check_case(k, seqs, fn, hd)
    except Exception as e:
        fails += 1; print("FAIL synthetic seed", seed, "k", k, repr(e)[:300])
        if fails > 5: break
    m += 1
    if time.time() - t0 > (float(sys.argv[1]) if len(sys.argv) > 1 else 240): break
print("gpu synthetic sets", m, "fails", fails, "time %.1f" % (time.time() - t0))
