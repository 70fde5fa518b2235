"""CPU restatement (TEST INFRASTRUCTURE ONLY -- never imported by simclr_amd/) of the two Keras optimizers that
/root/reference/tf2/model.py:31-34 instantiates for `--optimizer=momentum` / `--optimizer=adam`.  The arithmetic lives in
TensorFlow / Keras (not under /root/reference, not installable here: PARITY UNPINNED against TensorFlow itself); restated from
the published update rules of `tf.keras.optimizers.SGD` (ResourceApplyKerasMomentum) and `tf.keras.optimizers.Adam`
(ResourceApplyAdam), float64:

  SGD(lr, momentum, nesterov):  accum <- momentum * accum - lr * g
                                w     <- w + (momentum * accum - lr * g   if nesterov else   accum)
  Adam(lr, b1=0.9, b2=0.999, eps=1e-7), t = 1, 2, ...:
                                m <- b1 m + (1 - b1) g ;  v <- b2 v + (1 - b2) g^2
                                w <- w - lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)

`l2`: the gradient of the weight-decay LOSS term the reference adds for these optimizers (tf2/model.py:62-69:
weight_decay * sum_{non-BatchNorm v} l2_loss(v), l2_loss = sum(v^2) / 2), i.e. g + l2 * w.
"""
import numpy as np


def sgd_apply(w, g, accum, lr, momentum=0.0, nesterov=False, l2=0.0):
    w, g, accum = (np.asarray(a, np.float64) for a in (w, g, accum))
    g = g + l2 * w
    accum = momentum * accum - lr * g
    w = w + (momentum * accum - lr * g if nesterov else accum)
    return w, accum


def adam_apply(w, g, m, v, lr, t, beta_1=0.9, beta_2=0.999, epsilon=1e-7, l2=0.0):
    w, g, m, v = (np.asarray(a, np.float64) for a in (w, g, m, v))
    g = g + l2 * w
    # the hyper-parameters reach the TensorFlow kernel as float32 scalars and (1 - beta) is formed in float32 there: 1 - 0.999f =
    # 0.00099998713, 1.3e-5 off the real number -- restated, since it is a property of the reference's arithmetic, not a rounding of ours
    b1, b2 = float(np.float32(beta_1)), float(np.float32(beta_2))
    omb1, omb2 = float(np.float32(1.0) - np.float32(beta_1)), float(np.float32(1.0) - np.float32(beta_2))
    m = b1 * m + omb1 * g
    v = b2 * v + omb2 * g * g
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)        # beta_power = pow(float32(beta), t) in the TensorFlow kernel
    w = w - lr_t * m / (np.sqrt(v) + epsilon)
    return w, m, v
