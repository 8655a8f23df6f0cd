import torch
print("range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for p in (-1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "->", s.priority)
    except Exception as e:
        print(p, "err", e)
