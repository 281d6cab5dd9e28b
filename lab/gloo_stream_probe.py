"""Does gloo's CUDA all_reduce honour the current stream (a) for its input, (b) for its completion via work.wait()?"""
import os, sys, socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, port, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    torch.cuda.set_device(0)
    a = torch.randn(4096, 4096, device='cuda')
    s = torch.cuda.Stream()
    bad_in = bad_out = 0
    for it in range(20):
        x = torch.zeros(1 << 20, device='cuda')
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for _ in range(8):
                a = (a @ a).clamp(-1, 1)             # keep the stream busy for a while
            x.fill_(float(rank + 1))                   # produced late on stream s
            if mode == 'side':
                w = dist.all_reduce(x, async_op=True)
        if mode == 'main':
            torch.cuda.current_stream().wait_stream(s)
            w = dist.all_reduce(x, async_op=True)
        w.wait()
        y = x.clone()                                  # consumer on the current (default) stream
        torch.cuda.synchronize()
        v = float(y[0])
        if v != 3.0:
            bad_out += 1
    print(f'rank {rank} mode {mode}: wrong results {bad_out}/20 (last value {v})', flush=True)
    dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    for mode in ('main', 'side'):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context('spawn')
        ps = [ctx.Process(target=worker, args=(r, port, mode)) for r in range(2)]
        for p in ps: p.start()
        for p in ps: p.join()
