"""shared helpers for the -m gpu tests (they call the product only through the C-ABI binding)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pkg():
    return importlib.import_module("alphazero-risk_amd")


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
