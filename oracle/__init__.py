"""CPU oracle for the SimCLR pretraining hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy float64 / torch-CPU) of the arithmetic
in the reference's TF2 tree (`tf2/objective.py`, `tf2/lars_optimizer.py`,
`tf2/resnet.py`, `tf2/model.py`, `tf2/metrics.py`, `tf2/run.py:557-622`).
Every function cites the reference file:line it follows.

It is NOT part of the product.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it, and only as the checker.  The
product path (`simclr_amd/`) never imports `oracle` and fails loudly when the
HIP library is missing.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
this path and its arithmetic lives in TensorFlow, which
is not installed in this image (no network).  The oracle is therefore pinned
only by (a) closed-form known answers (tests/test_oracle_*.py), (b) the
structural known answers the reference does publish (parameter counts in
README.md:19-33, endpoint shapes in tf2/colabs/finetuning.ipynb:909), and
(c) float64-vs-autograd self-consistency.  If `import tensorflow` ever
succeeds, `oracle/check_against_tf.py` runs /root/reference/tf2 directly.
"""
