"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's meta-ASR hot path,
written from the reference's behaviour (file:line cited per function) and
pinned against golden vectors captured from the real reference running in the
build container (`oracle/make_goldens.py` -> `tests/golden/*.npz`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package, and only as the checker / reported baseline.  The
product path (`metaasr-crossaccent_amd/`) never imports it and has no CPU
fallback: it raises when the HIP library is missing.

Parity status: PINNED by captured goldens (torch 2.10.0 CPU running the
reference's own `MyTransformer`, `run_batch`, `FOMetaASRInterface`,
`BucketSampler`, `TransformerOptimizer`).  The reference ships no tests or
fixtures of its own, so nothing else pins it.
Reptile has no reference implementation: parity unpinned there.
"""
