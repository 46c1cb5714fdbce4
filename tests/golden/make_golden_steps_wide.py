"""Round-6 golden vectors from the GENUINE reference (build container only; see make_golden.py / ref_import.py):

  srflow_steps_wide.npz    two CONSECUTIVE coupled FlowSteps of level 3 (layers 42, 43: C = 96 flow channels, conditional = the 320 stacked
                           feature channels at that level's resolution), forward and reverse, B = 2 at 10 x 36 (two 8-row x two 32-pixel tiles,
                           ragged both ways) -- the inputs the engine's level-3 hot path needs: the batched 320 -> 16*64 hoists, the 1x1-only head
                           + Conv2dZeros of fFeatures, then coupling_wide_head -> coupling_wide_tail (coupling_wide.hip; FlowStep.py:88-129,
                           FlowAffineCouplingsAblation.py:57-135).
Usage:  python tests/golden/make_golden_steps_wide.py
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402
from make_golden import maxdiff, rnd, save  # noqa: E402

torch.set_grad_enabled(False)


def main():
    from bfsr_amd import synth
    from bfsr_amd.srflow import options, spec
    import oracle.srflow_ref as O
    opt_ref = R.srflow_opt(4)
    opt = options.load(options.DEFAULT_CONF)
    net = R.build_srflownet(opt_ref)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    net.load_state_dict(sd, strict=True)
    out, man = OrderedDict(), {}
    B, C, H, W, la, lb = 2, 96, 10, 36, 42, 43
    z = rnd(600, B, C, H, W)
    ft = rnd(620, B, 320, H, W, scale=0.5)
    out["l3_ft"] = ft
    ma, mb = net.flowUpsamplerNet.layers[la], net.flowUpsamplerNet.layers[lb]
    ld = torch.zeros(B)
    fa, _ = ma(z, ld, reverse=False, rrdbResults=ft)                 # encode order: a then b
    fab, _ = mb(fa, ld, reverse=False, rrdbResults=ft)
    rb, _ = mb(z, ld, reverse=True, rrdbResults=ft)                  # decode order: b then a
    rba, _ = ma(rb, ld, reverse=True, rrdbResults=ft)
    out.update({"l3_z": z, "l3_fwd_a": fa, "l3_fwd_ab": fab, "l3_rev_b": rb, "l3_rev_ba": rba})
    pa, pb = "flowUpsamplerNet.layers.%d" % la, "flowUpsamplerNet.layers.%d" % lb
    ofa = O.flow_step(z, ft, sd, pa, True, False)
    orb = O.flow_step(z, ft, sd, pb, True, True)
    man["l3"] = max(maxdiff(ofa, fa), maxdiff(O.flow_step(ofa, ft, sd, pb, True, False), fab),
                    maxdiff(orb, rb), maxdiff(O.flow_step(orb, ft, sd, pa, True, True), rba))
    out["weights_seed"] = np.int64(1234)
    out["weights_sha256"] = np.frombuffer(synth.digest(sd).encode(), dtype=np.uint8)
    save("srflow_steps_wide.npz", **out)
    path = os.path.join(HERE, "MANIFEST.json")
    full = json.load(open(path))
    full.setdefault("srflow", {})["flowstep_pairs_wide"] = man
    json.dump(full, open(path, "w"), indent=1, sort_keys=True)
    print("oracle vs reference:", man)


if __name__ == "__main__":
    main()
