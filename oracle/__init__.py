"""CPU oracle for the OccuSeg sparse-conv hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in numpy, the algorithm of the reference's GPU path
(site ordering, rulebooks) and CPU arithmetic (gather -> mm -> scatter, batch
norm, input/output layers).  It exists so that tests can check the HIP path
against an independent CPU computation.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``occuseg_amd/`` imports it: the product path
has no CPU fallback and fails loudly when the HIP library is missing.

Parity pin status
-----------------
The reference ships no tests, fixtures or golden vectors for this path
 and its extension cannot be built here (it needs CUDA 9 +
CUDPP; the CPU-only translation unit is stale).  Therefore:

* rulebook / site-order semantics (``oracle.rules``): **parity unpinned** by
  reference outputs.  They are pinned only by (i) a second, independent
  dict-based restatement (``oracle.bruteforce``) and (ii) the structural
  invariants the reference's CUDA code implies.
* arithmetic (``oracle.arith``): pinned against the reference's own CPU loops
  (``SCN/CPU/BatchNormalization.cpp``, ``SCN/CPU/IOLayers.cpp`` and the
  ``rule_index_select`` / ``rule_index_add_`` helpers of
  ``SCN/CPU/Convolution.cpp``) compiled from where they lie into
  ``oracle/_ref/`` (see ``oracle/Makefile``); golden vectors produced by that
  build are committed under ``tests/golden/``.
"""
