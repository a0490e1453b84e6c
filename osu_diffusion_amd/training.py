"""Training path (autograd bridge, fused loss, AdamW+EMA, gradient all-reduce).  Filled in by the
training milestone; until then a gradient-requiring forward fails loudly instead of falling back."""
from . import _lib


def dit_forward_autograd(model, x, t, o, c, y, attn_mask):
    model.native_handle()  # raises if the module is not on a GPU
    raise _lib.NativeError("the native backward pass is not built in this revision; wrap inference in torch.no_grad()")
