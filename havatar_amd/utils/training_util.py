"""The helpers of the reference's utils/training_util.py that the hot path and its harnesses use (:24-39, :124-139)."""
import math


def mse2psnr(mse):
    return -10.0 * math.log10(1e-5 if mse == 0 else mse)


def get_minibatches(inputs, chunksize=1024 * 8, dim=0):
    """list of views of `inputs`, `chunksize` long along dim 0 (or dim 1 if dim != 0) (reference :31-39)."""
    if dim == 0:
        return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]
    return [inputs[:, i:i + chunksize] for i in range(0, inputs.shape[1], chunksize)]


def load_partial_state_dict(model, loaded_dict, except_keys, full_name=False):
    """load_state_dict skipping keys in (or, unless full_name, prefixed by) except_keys (reference :124-139)."""
    if full_name:
        keep = {k: v for k, v in loaded_dict.items() if k not in except_keys}
    else:
        keep = {k: v for k, v in loaded_dict.items() if not any(k.startswith(e) for e in except_keys)}
    sd = model.state_dict()
    sd.update(keep)
    model.load_state_dict(sd)
