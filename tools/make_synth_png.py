"""Write the synthetic 256x256 inputs of BASELINE config 1: uniform noise, seeds 7 and 8."""
import os
import torch
from PIL import Image
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for seed, fn in ((7, "synth_content_256.png"), (8, "synth_style_256.png")):
    g = torch.Generator().manual_seed(seed)
    arr = (torch.rand(256, 256, 3, generator=g) * 255).byte().numpy()
    Image.fromarray(arr).save(os.path.join(REPO, "tests", fn))
    print("wrote", fn)
