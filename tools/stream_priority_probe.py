#!/usr/bin/env python3
"""HIP stream priority range as PyTorch exposes it (here: 0 = least, -1 = greatest; the default stream is 0)."""
import torch
print("range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for p in (-1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "->", s.priority)
    except Exception as e:
        print(p, "err", e)
