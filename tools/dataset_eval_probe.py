import sys, json, argparse
sys.path.insert(0, '.')
import torch, bench
a = bench.parse([])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
which = sys.argv[1]
if which == "after_ucf":
    from iefvad_amd import synth
    margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
    sd = synth.make_state_dict(0, 768, 2, 10)
    bench.ucf_eval(sd, margs, dev, a)
r = bench.dataset_eval("c3", bench.xd_parts(), 17, 10, "bf16", dev, a, batch_chunks=128, lanes=2)
print(which, "xd", r["snippets_per_s"], r["seconds_all_passes"])
r = bench.dataset_eval("c5", bench.config5_parts(), 19, 5, "bf16", dev, a, batch_chunks=128, lanes=2)
print(which, "c5", r["snippets_per_s"], r["seconds_all_passes"])
