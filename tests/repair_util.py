"""sequence_end_repair on the device (ac_end_repair_device) against the oracle's restatement of compress.rs:202-270."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as O
from autocycler_amd import _capi


def check_repair(lib_path, k, seqs, filenames, headers, device="cpu", device_index=0):
    lib = _capi.load_library(lib_path)
    raw = O.Seqs.from_raw(k, seqs, filenames=filenames, headers=headers, repair=False).all()
    want = O.Seqs.from_raw(k, seqs, filenames=filenames, headers=headers, repair=True).all()
    n = len(raw)
    views = (_capi.SeqView * n)()
    keep = []
    for i, q in enumerate(raw):
        b = bytes(q["fwd"]); keep.append(b)
        views[i].fwd, views[i].length, views[i].id = b, q["length"], q["id"]
    n_text = lib.ac_text_size(C.c_uint32(k), views, C.c_uint32(n))
    text = np.empty(n_text, dtype=np.uint8)
    off = (C.c_uint64 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
    assert lib.ac_layout_text(C.c_uint32(k), views, C.c_uint32(n), text.ctypes.data_as(C.c_void_p), off, d1, d2) == 0
    lens = (C.c_uint32 * n)(*[q["length"] for q in raw])
    d_text = torch.from_numpy(text).to(device)
    secs, nm = C.c_double(), C.c_uint64()
    rc = lib.ac_end_repair_device(C.c_uint32(k), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text), off, lens, d1, d2, C.c_uint32(n),
                                  C.c_int(device_index), C.byref(secs), C.byref(nm))
    assert rc == 0, lib.ac_last_error()
    got = d_text.cpu().numpy().tobytes()
    for i, q in enumerate(want):
        plen = q["length"] + k - 1
        g = got[off[i]:off[i] + plen]
        assert g == q["fwd"], f"sequence {i}: oracle={q['fwd'][:k + 5]!r}..{q['fwd'][-k - 5:]!r} got={g[:k + 5]!r}..{g[-k - 5:]!r}"
        lead = len(g) - len(g.lstrip(b"."))
        trail = len(g) - len(g.rstrip(b"."))
        assert (d1[i], d2[i]) == (lead, trail)
    return d_text, off, lens, d1, d2, nm.value, secs.value
