import torch
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p)
        print('priority', p, 'ok ->', s.priority)
    except Exception as e:
        print('priority', p, 'rejected:', str(e)[:80])
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'no priority_range')
