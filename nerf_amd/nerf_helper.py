"""Host-side mirror of the reference's ``nerf/nerf_helper.py`` (same names and argument meaning)."""
import torch

from . import ops


def positional_encoding(x: torch.Tensor, freq_level: int) -> torch.Tensor:
    """[sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] on the last dim -- HIP kernel (nerf_helper.py:38-48)."""
    return ops.positional_encoding(x, freq_level)


def makeMLP(in_chan, out_chan, act=torch.nn.ReLU(), batch_norm=False):
    """Linear (+BatchNorm1d) (+activation) as a list of modules (nerf_helper.py:17-23)."""
    layers = [torch.nn.Linear(in_chan, out_chan)]
    if batch_norm:
        layers.append(torch.nn.BatchNorm1d(out_chan))
    if act is not None:
        layers.append(act)
    return layers


def saveModel(model, path: str, other_stuff: dict = None, opt=None, amp=None):
    """Checkpoint dict {'model', 'optimizer'?, 'amp'?, ...} (nerf_helper.py:7-15)."""
    ckpt = {"model": model.state_dict()}
    if amp is not None:
        ckpt["amp"] = amp.state_dict()
    if opt is not None:
        ckpt["optimizer"] = opt.state_dict()
    if other_stuff is not None:
        ckpt.update(other_stuff)
    torch.save(ckpt, path)


def nan_hook(self, inp, output):
    """Forward hook raising on NaN outputs (nerf_helper.py:26-36)."""
    outs = output if isinstance(output, tuple) else (output,)
    for i, o in enumerate(outs):
        bad = torch.isnan(o)
        if bad.any():
            raise RuntimeError("Found NAN in output %d of %s at %s" % (i, self.__class__.__name__, bad.nonzero()))


def linear_to_srgb(linear: torch.Tensor, eps: float = None) -> torch.Tensor:
    """sRGB transfer curve (nerf_helper.py:50-56)."""
    if eps is None:
        eps = torch.full((1,), torch.finfo(torch.float32).eps, device=linear.device)
    low = 323 / 25 * linear
    high = (211 * torch.maximum(eps, linear) ** (5 / 12) - 11) / 200
    return torch.where(linear <= 0.0031308, low, high)
