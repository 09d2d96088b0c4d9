#!/usr/bin/env python3
"""Text bytes of a named workload in the layout of include/autocycler_hip.h ('$' + padded sequences each followed by '$'): what
tools/pmc_traffic.py calibrates PackFunctor's bytes against.   python tools/workload_n_text.py configEprime_k51"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from autocycler_amd import synth

k, _, gen = synth.WORKLOADS[sys.argv[1]]
print(1 + sum(len(s) + (k - 1) + 1 for contigs in gen() for _, s in contigs))
