"""CPU oracle for the SimCLR pretraining hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy float64 / torch-CPU) of the arithmetic
in the reference's TF2 tree (`tf2/objective.py`, `tf2/lars_optimizer.py`,
`tf2/resnet.py`, `tf2/model.py`, `tf2/metrics.py`, `tf2/run.py:557-622`).
Every function cites the reference file:line it follows.

It is NOT part of the product.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it, and only as the checker.  The
product path (`simclr_amd/`) never imports `oracle` and fails loudly when the
HIP library is missing.

PARITY STATUS -- pinned to the reference's SOURCE, not to TensorFlow's kernels.  The reference ships no tests, golden
vectors or fixtures for this path and its arithmetic sits on TensorFlow, which cannot be installed in
this image (no network).  Since round 4 the reference's own files -- /root/reference/tf2/{objective,lars_optimizer,metrics,
resnet,data_util,model}.py, unmodified -- are executed in the build container on top of `oracle/tfshim.py`, a float64 numpy
stand-in for the TensorFlow / Keras / absl calls they make, and their outputs are committed as tests/golden/reference_pin.npz
(script: tests/golden/make_reference_golden.py; 265 arrays).  tests/test_reference_pin.py requires every one of them from the
oracle: NT-Xent loss / logits / labels for one and for R emulated replicas (+ the loss gradient by central differences of the
reference's function), the supervised loss, every LARS branch and name filter over two steps, the learning-rate schedule, the
weight decay, the metrics, the blur filter, FixedPadding + Conv2dFixedPadding, BatchNormRelu (training / moving averages /
inference), the composition of the two-view augmentation with scripted draws (data_util.preprocess_image: gates, order, clips,
ranges; NOT its pixel kernels), the training step `single_step` of tf2/run.py:557-622 (compiled from main()'s own source; loss
composition, 1/R, metrics, on 1 and 2 emulated replicas), and the whole `Model` forward (ResNet-18 CIFAR stem, ResNet-50, ResNet-50 + SK / ResNet-D,
ResNet-34 x2 with a two-layer head, local BatchNorm without the linear-eval head) with its variable NAMES, shapes, trainability
and initial values -- agreement 1e-10 or better in float64.  What this does NOT pin is
TensorFlow itself: the primitives under the reference's code (conv2d SAME/VALID, Keras BatchNormalization with the biased
variance, avg-pool SAME counting valid elements, CosineDecay, softmax cross-entropy, l2_normalize's epsilon) are textbook
numpy following the documented semantics, each named in tfshim.py.  `oracle/check_against_tf.py` still runs the same
comparisons against a real TensorFlow wherever one is importable.  oracle/augment.py (four TensorFlow image kernels that are
not under /root/reference) stays PARITY UNPINNED.
"""
