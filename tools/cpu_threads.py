# scratch: how many host threads give the oracle its best frames/s on this box
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import generator as og
from animateportrait_amd.synthetic import make_generator_inputs, generator_args
sd = og.init_params(og.generator_param_shapes(3, 1, 64, 9, 3, 3), seed=1234)
args = generator_args(make_generator_inputs(4, seed=1234))
for th in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(th)
    with torch.no_grad():
        og.generator_forward(sd, *args, div=3, disp=3)
        t0 = time.time(); og.generator_forward(sd, *args, div=3, disp=3); dt = time.time() - t0
    print('threads %d: %.2f frames/s' % (th, 4 / dt), flush=True)
