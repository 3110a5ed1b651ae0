"""Device pick (reference: utils/device.py:4-10).  Under one-process-per-GPU data parallelism the
decoders allocate on their input's device instead of this import-time global."""
import torch


def get_device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")
