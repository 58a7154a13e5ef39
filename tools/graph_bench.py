#!/usr/bin/env python3
"""Per-frame generator forward at small batch, eager vs replayed as a HIP graph (torch.cuda.CUDAGraph capture of the C-ABI
launches on the capture stream).  Usage: python tools/graph_bench.py [batch ...]"""
import contextlib
import io
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animateportrait_amd import networks
from animateportrait_amd.synthetic import make_generator_inputs, generator_args


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    batches = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    dev = torch.device('cuda:0')
    with contextlib.redirect_stdout(io.StringIO()):
        G = networks.define_G(3, 1, 64, 'resnet_9blocks_rcatland32_full_ifw', 'instance', False, 'normal', 0.02, [0],
                              div=3, disp=3)
    for n in batches:
        args = [t.to(dev) for t in generator_args(make_generator_inputs(n, seed=1))]
        with torch.no_grad():
            y_ref = G(*args)
            t_eager = timed(lambda: G(*args), 50)
            static_in = [a.clone() for a in args]
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    G(*static_in)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y_static = G(*static_in)

            def replay():
                for d, a in zip(static_in, args):
                    d.copy_(a)
                graph.replay()
            replay()
            torch.cuda.synchronize()
            err = float((y_static - y_ref).abs().max())
            t_graph = timed(replay, 200)
        print('B=%d  eager %.3f ms (%.0f frames/s)   graph replay %.3f ms (%.0f frames/s)   max |diff| %.1e' % (
            n, t_eager * 1e3, n / t_eager, t_graph * 1e3, n / t_graph, err), flush=True)


if __name__ == '__main__':
    main()
