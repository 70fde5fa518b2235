"""Oracle: LARS update, LR schedule, weight-decay term (numpy float64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED TO THE REFERENCE'S SOURCE (tests/golden/reference_pin.npz: tf2/*.py executed on oracle/tfshim.py); TensorFlow's own kernels unpinned.

Restates /root/reference/tf2/lars_optimizer.py:83-157 and
/root/reference/tf2/model.py:47-116.
"""
import math
import re

import numpy as np

EETA_DEFAULT = 0.001  # tf2/lars_optimizer.py:22


def use_weight_decay(name, weight_decay, exclude_from_weight_decay):
    """tf2/lars_optimizer.py:139-148."""
    if not weight_decay:
        return False
    if exclude_from_weight_decay:
        for r in exclude_from_weight_decay:
            if re.search(r, name) is not None:
                return False
    return True


def do_layer_adaptation(name, exclude_from_layer_adaptation):
    """tf2/lars_optimizer.py:150-157."""
    if exclude_from_layer_adaptation:
        for r in exclude_from_layer_adaptation:
            if re.search(r, name) is not None:
                return False
    return True


def lars_apply(name, param, grad, v, learning_rate, momentum=0.9, use_nesterov=False,
               weight_decay=0.0, exclude_from_weight_decay=None,
               exclude_from_layer_adaptation=None, classic_momentum=True,
               eeta=EETA_DEFAULT, dtype=np.float64):
    """One `_resource_apply_dense` (tf2/lars_optimizer.py:83-137).

    Returns (next_param, next_v).  `dtype` float32 reproduces the reference's
    fp32 arithmetic order; float64 is the ground truth.
    """
    # :72-77: exclude_from_layer_adaptation defaults to exclude_from_weight_decay
    if not exclude_from_layer_adaptation:
        exclude_from_layer_adaptation = exclude_from_weight_decay
    param = np.asarray(param, dtype=dtype)
    grad = np.asarray(grad, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    lr = dtype(learning_rate)
    if use_weight_decay(name, weight_decay, exclude_from_weight_decay):
        grad = grad + dtype(weight_decay) * param                       # :96-97
    if classic_momentum:                                                # :99
        trust_ratio = dtype(1.0)
        if do_layer_adaptation(name, exclude_from_layer_adaptation):    # :101
            w_norm = np.sqrt(np.sum(param * param, dtype=dtype))        # :102
            g_norm = np.sqrt(np.sum(grad * grad, dtype=dtype))          # :103
            if w_norm > 0 and g_norm > 0:                               # :104-107
                trust_ratio = dtype(eeta) * w_norm / g_norm
        scaled_lr = lr * trust_ratio                                    # :108
        next_v = dtype(momentum) * v + scaled_lr * grad                 # :110
        update = dtype(momentum) * next_v + scaled_lr * grad if use_nesterov else next_v  # :111-114
        next_param = param - update                                     # :115
    else:
        next_v = dtype(momentum) * v + grad                             # :117
        update = dtype(momentum) * next_v + grad if use_nesterov else next_v   # :118-121
        trust_ratio = dtype(1.0)
        if do_layer_adaptation(name, exclude_from_layer_adaptation):    # :124
            w_norm = np.sqrt(np.sum(param * param, dtype=dtype))
            v_norm = np.sqrt(np.sum(update * update, dtype=dtype))
            if w_norm > 0 and v_norm > 0:                               # :127-130
                trust_ratio = dtype(eeta) * w_norm / v_norm
        scaled_lr = trust_ratio * lr                                    # :131
        next_param = param - scaled_lr * update                         # :132
    return next_param, next_v


def get_train_steps(num_examples, train_steps, train_epochs, train_batch_size):
    """tf2/model.py:72-75."""
    return train_steps or (num_examples * train_epochs // train_batch_size + 1)


def warmup_and_cosine_decay(step, base_learning_rate, num_examples, *, warmup_epochs=10,
                            train_batch_size=512, learning_rate_scaling='linear',
                            train_epochs=100, train_steps=0):
    """WarmUpAndCosineDecay.__call__ (tf2/model.py:87-110).

    tf.keras.experimental.CosineDecay(lr, decay_steps)(s) =
    lr * 0.5 * (1 + cos(pi * min(s, decay_steps) / decay_steps))  (alpha=0).
    """
    warmup_steps = int(round(warmup_epochs * num_examples // train_batch_size))   # :89-91
    if learning_rate_scaling == 'linear':
        scaled_lr = base_learning_rate * train_batch_size / 256.                  # :92-93
    elif learning_rate_scaling == 'sqrt':
        scaled_lr = base_learning_rate * math.sqrt(train_batch_size)              # :94-95
    else:
        raise ValueError('Unknown learning rate scaling {}'.format(learning_rate_scaling))
    learning_rate = step / float(warmup_steps) * scaled_lr if warmup_steps else scaled_lr  # :99-100
    total_steps = get_train_steps(num_examples, train_steps, train_epochs, train_batch_size)  # :103
    decay_steps = total_steps - warmup_steps                                       # :105-106
    s = min(max(step - warmup_steps, 0), decay_steps)
    cosine = scaled_lr * 0.5 * (1.0 + math.cos(math.pi * s / decay_steps))
    return learning_rate if step < warmup_steps else cosine                        # :107-108


def add_weight_decay_lars(named_params, weight_decay):
    """tf2/model.py:47-60 (adjust_per_optimizer and 'lars'): only the supervised
    head's non-bias variables; tf.nn.l2_loss(v) = sum(v**2)/2."""
    l2 = [0.5 * float(np.sum(np.asarray(p, dtype=np.float64) ** 2))
          for name, p in named_params
          if 'head_supervised' in name and 'bias' not in name]
    return weight_decay * sum(l2) if l2 else 0.0
