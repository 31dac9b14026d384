"""Model files, as the reference writes them (jova/utils/io.py:314-361): ``<path>/<name>.mod`` holds the model's
state dict saved with torch.save, so files move freely between the reference and this package (the modules here keep the
reference's state-dict keys and shapes)."""
import os

import torch


def save_model(model, path, name):
    """Saves the model parameters (and buffers) to ``<path>/<name>.mod`` (io.py:314-325)."""
    os.makedirs(path, exist_ok=True)
    file = os.path.join(path, name + ".mod")
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, file)
    return file


def load_model(path, name, dvc=None):
    """Loads a saved state dict from ``<path>/<name>`` (io.py:350-361; ``name`` carries the ".mod" suffix, as in the
    reference's calls).  The file is read with ``weights_only=True``: tensors only, nothing in it is executed."""
    if dvc is None:
        dvc = torch.device("cuda:0")
    return torch.load(os.path.join(path, name), map_location=dvc, weights_only=True)
