#!/usr/bin/env python
"""README embedding example of the reference (scripts/embed.py): embed random audio with AFx-Rep."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from st_ito.utils import get_param_embeds, load_param_model, make_synthetic_param_model  # noqa: E402

if __name__ == "__main__":
    use_gpu = torch.cuda.is_available()
    try:
        model = load_param_model(use_gpu=use_gpu)
    except FileNotFoundError as e:
        print(f"{e}\n-> falling back to seeded random weights for this demo")
        model = make_synthetic_param_model()
    x = torch.randn(1, 2, 262144)
    embeds = get_param_embeds(x, model, 48000)
    for k, v in embeds.items():
        print(k, tuple(v.shape))
