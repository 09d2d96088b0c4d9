"""One rank of a multi-process sharded build (launched by the tests, one process per rank; gloo on CPU, or two ranks
sharing the one GPU of the test box with gloo staging).  Usage:
    python sharded_worker.py RANK WORLD PORT LIB_PATH DEVICE CASES [BACKEND]     CASES = "k:seed,k:seed,...", "synth:k", "mixed:k" or "big:assemblies:genome"
BACKEND (default gloo): "nccl" = RCCL; with WORLD = 1 every collective is still issued (Comm(always_collective=True)), which is how the
one-GPU test box exercises the RCCL calls of autocycler_amd/sharded.py on device tensors.
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def big_case(lib_path, n_asm, genome, dev, rank, world):
    import ctypes as C
    import numpy as np
    import torch.distributed as dist
    import sharded_util
    from autocycler_amd import _capi, compress_build, sharded, synth
    k = 51
    lib = _capi.load_library(lib_path)
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(synth.make_assemblies(n_asm, genome=genome, plasmid=genome // 50, sub=2e-4, indel=2e-5, seed=31337)):
        for header, s in contigs:
            seqs.append(np.ascontiguousarray(s)); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    n = len(seqs)
    out = C.c_void_p()
    rc = lib.ac_seqs_from_raw(C.c_uint32(k), C.c_uint32(n), (C.c_void_p * n)(*[s.ctypes.data for s in seqs]), (C.c_uint32 * n)(*[len(s) for s in seqs]),
                              (C.c_char_p * n)(*[x.encode() for x in fn]), (C.c_char_p * n)(*[x.encode() for x in hd]), C.c_uint32(n_asm), C.c_int(1),
                              C.c_int(8), C.byref(out))
    assert rc == 0, lib.ac_last_error()
    views = lib.ac_seqs_views(out)
    loaded = [dict(fwd=C.string_at(views[i].fwd, views[i].length + k - 1), length=views[i].length, id=views[i].id) for i in range(n)]
    lib.ac_seqs_free(out)
    b = sharded_util.slice_bounds(n, world)
    shard = sharded_util.local_shard(lib, k, loaded, b[rank], b[rank + 1], max(1, n_asm // world), dev)
    g, info = sharded.sharded_build(lib, shard, sharded.Comm(dev), device_index=dev.index or 0, root=0, gather_paths=True)
    if rank == 0:
        gfa_sharded = g.gfa(fn, hd)
        single = compress_build(k, n_asm, [(q["fwd"], q["length"], q["id"]) for q in loaded], device=dev.index or 0, lib_path=lib_path)
        gfa_single = single.gfa(fn, hd)
        assert g.stats_post == single.stats_post and g.kmer_count == single.kmer_count
        assert gfa_sharded == gfa_single, "sharded and single-device GFA differ"
        assert info["fragments"] > 2 * n and g.stats_post["unitigs"] > 10
        if os.environ.get("AC_PATH_COPY") == "1" and os.environ.get("AC_SHARD_PATH_COPY") != "0":      # (test_copying_walk_over_gloo)
            assert info["path_runs_copied_here"] > 0, info
        print(f"big case: {n} sequences, {g.stats_post['unitigs']} unitigs, {info['fragments']} fragments, union text {info['union_text_bytes']} bytes")
    dist.barrier()


def partition_case(lib_path, dev, rank, world):
    """The k-mer table is PARTITIONED over the ranks (owner = hash of the canonical middle): on a mixed-species job — one species
    per rank, bench.py's N > 1 workload in miniature — every rank's share of the table has about 1/world of the slots the
    single-device build of the same job needs, the shares of the novel bitmap are disjoint, and the result is still the oracle's."""
    import torch
    import torch.distributed as dist
    import sharded_util
    from autocycler_amd import compress_build, sharded, synth
    import oracle_lib as O
    k = 51
    seqs, fn, hd = [], [], []
    for sp in range(world):
        for i, contigs in enumerate(synth.make_assemblies(2, genome=60_000, plasmid=3_000, sub=1e-3, indel=1e-4, seed=100 + 1000 * sp)):
            for header, s in contigs:
                seqs.append(s.tobytes().decode()); fn.append(f"assembly_{2 * sp + i:04d}.fasta"); hd.append(header)
    comm = sharded.Comm(dev)
    lib = sharded_util._capi.load_library(lib_path)
    s_all = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd, repair=True)
    loaded = s_all.all()
    b = sharded_util.slice_bounds(len(loaded), world)
    shard = sharded_util.local_shard(lib, k, loaded, b[rank], b[rank + 1], 2, dev)
    g, info = sharded.sharded_build(lib, shard, comm, device_index=dev.index or 0, root=0, gather_paths=True)
    caps = torch.tensor([info["table_capacity"]], dtype=torch.int64)
    all_caps = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_caps, caps)
    if rank == 0:
        gfa_o, st, _ = s_all.compress(k)
        assert g.gfa(fn, hd) == gfa_o, "partitioned build differs from the oracle"
        single = compress_build(k, 2 * world, [(q["fwd"], q["length"], q["id"]) for q in loaded], device=dev.index or 0, lib_path=lib_path)
        single_cap = single.timings()["table_capacity"]
        per_rank = [int(c.item()) for c in all_caps]
        distinct = single.timings()["n_distinct"]
        # a share holds ~distinct / world keys; capacities are powers of two, so allow a factor of two around the ideal share
        assert max(per_rank) * world <= 2 * single_cap, (per_rank, single_cap)
        assert min(per_rank) * world * 4 >= single_cap, (per_rank, single_cap)
        assert max(per_rank) >= distinct // world, (per_rank, distinct)          # ... and still holds its share of the keys
        print(f"partition case: world {world}, single-device table {single_cap} slots for {distinct} k-mers, per-rank shares {per_rank}")
    dist.barrier()


def main():
    rank, world, port, lib_path, device, cases = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    backend = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    if device == "cuda:rank":      # one distinct GPU per rank (a multi-GPU box)
        device = f"cuda:{rank}"
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(torch.device(device))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(device))
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import seqgen
    import sharded_util
    from autocycler_amd import sharded
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    done = 0
    for case in cases.split(","):
        if case == "partition":
            partition_case(lib_path, dev, rank, world)
            done += 1
            continue
        if case == "badinput":
            # the LAST rank's slice holds a foreign byte: ac_shard_begin refuses it there — and every rank must raise, none may be left waiting in
            # the first collective (the status travels with the fragment sizes: autocycler_amd/sharded.py)
            from autocycler_amd import AutocyclerError
            import oracle_lib as O
            seqs, fn, hd = seqgen.make_case(3, 11)
            loaded = O.Seqs.from_raw(11, seqs, filenames=fn, headers=hd).all()
            b = sharded_util.slice_bounds(len(loaded), world)
            part = [dict(q) for q in loaded[b[rank]:b[rank + 1]]]
            if rank == world - 1:
                fwd = bytearray(part[0]["fwd"]); fwd[len(fwd) // 2] = ord("N"); part[0]["fwd"] = bytes(fwd)
            lib = sharded_util._capi.load_library(lib_path)
            shard = sharded_util.local_shard(lib, 11, part, 0, len(part), 1, dev)
            try:
                sharded.sharded_build(lib, shard, sharded.Comm(dev), device_index=dev.index or 0)
                raise SystemExit("a sharded build with a foreign byte in one slice went through")
            except AutocyclerError as e:
                msg = str(e)
                assert ("non-ACGT" in msg) if rank == world - 1 else (f"rank {world - 1}" in msg), msg
            # ... and the library is usable again on every rank
            comm = sharded.Comm(dev)
            sharded_util.run_case(lib_path, 11, seqs, fn, hd, comm, dev)
            done += 1
            continue
        if case.startswith("big:"):
            # a realistic-size job (too big for the oracle): the sharded result must equal the single-device build of the same
            # sequences, which the full-size property tests pin (tests/test_gpu_fullsize.py)
            _, n_asm, genome = case.split(":")
            big_case(lib_path, int(n_asm), int(genome), dev, rank, world)
            done += 1
            continue
        a, b = case.split(":")
        if a == "synth":
            from autocycler_amd import synth
            k = int(b)
            seqs, fn, hd = [], [], []
            for i, contigs in enumerate(synth.make_assemblies(6, genome=40_000, plasmid=2_000, sub=1e-3, indel=1e-4, seed=77)):
                for header, s in contigs:
                    seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
        elif a == "mixed":      # a mixed-species job (bench.py's N > 1 workload in miniature): 3 species x 3 assemblies
            from autocycler_amd import synth
            k = int(b)
            seqs, fn, hd = [], [], []
            n_species, per = (world, 2) if world > 3 else (3, 3)      # bench.py's layout: one species per rank
            for sp in range(n_species):
                for i, contigs in enumerate(synth.make_assemblies(per, genome=30_000, plasmid=1_500, sub=1e-3, indel=1e-4, seed=100 + 1000 * sp)):
                    for header, s in contigs:
                        seqs.append(s.tobytes().decode()); fn.append(f"assembly_{per * sp + i:04d}.fasta"); hd.append(header)
        else:
            k, seed = int(a), int(b)
            seqs, fn, hd = seqgen.make_case(seed, k)
        if len(seqs) < world:
            continue
        for repair, gather in ((True, True), (False, False)):
            comm = sharded.Comm(dev, always_collective=(backend == "nccl"))
            sharded_util.run_case(lib_path, k, seqs, fn, hd, comm, dev, repair=repair, device_index=dev.index or 0,
                                  gather_paths=gather)
            if backend == "nccl":
                for name in ("all_gather_into_tensor", "all_to_all_single", "all_reduce_SUM", "all_reduce_MIN") + (("gather",) if gather else ()):
                    assert comm.calls.get(name, 0) > 0, f"collective {name} was not issued"
        done += 1
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {done} cases OK ({backend})")


if __name__ == "__main__":
    main()
