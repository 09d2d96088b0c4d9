"""One rank of a multi-process sharded build (launched by the tests, one process per rank; gloo on CPU, or two ranks
sharing the one GPU of the test box with gloo staging).  Usage:
    python sharded_worker.py RANK WORLD PORT LIB_PATH DEVICE CASES     CASES = "k:seed,k:seed,..." or "synth:k"
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def main():
    rank, world, port, lib_path, device, cases = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import seqgen
    import sharded_util
    from autocycler_amd import sharded
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    done = 0
    for case in cases.split(","):
        a, b = case.split(":")
        if a == "synth":
            from autocycler_amd import synth
            k = int(b)
            seqs, fn, hd = [], [], []
            for i, contigs in enumerate(synth.make_assemblies(6, genome=40_000, plasmid=2_000, sub=1e-3, indel=1e-4, seed=77)):
                for header, s in contigs:
                    seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
        else:
            k, seed = int(a), int(b)
            seqs, fn, hd = seqgen.make_case(seed, k)
        if len(seqs) < world:
            continue
        for repair, gather in ((True, True), (False, False)):
            comm = sharded.Comm(dev)
            sharded_util.run_case(lib_path, k, seqs, fn, hd, comm, dev, repair=repair, device_index=dev.index or 0,
                                  gather_paths=gather)
        done += 1
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {done} cases OK")


if __name__ == "__main__":
    main()
