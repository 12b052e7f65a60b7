"""Training (autograd) entry points of the HIP path.

Round 1 ships the forward (sampling / evaluation) kernels; the backward kernels (conv dgrad/wgrad, GroupNorm-Swish
backward, flash-attention backward, embedding/linear backward) are the next row of the scope table (SURVEY.md section 8,
configs C3/C4).  Until they exist the training entry points refuse loudly instead of falling back to another backend.
"""


def unet_forward_with_grad(model, x, t, labels):
    raise NotImplementedError(
        "hdiff: the backward kernels of the HIP path are not built yet; call the model under torch.no_grad() "
        "(sampling / evaluation).  There is deliberately no fallback to another backend.")


def sq_err_with_grad(eps_hat, noise):
    raise NotImplementedError("hdiff: backward kernels not built yet (see autograd.unet_forward_with_grad)")
