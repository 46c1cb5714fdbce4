"""Round-5 golden vectors from the GENUINE reference (build container only; see make_golden.py / ref_import.py):

  srflow_steps_fused.npz   two CONSECUTIVE coupled FlowSteps of level 1 (layers 3, 4: C = 12, conditional = cat[fea_up2, nearest-x2 of the
                           LR-resolution block taps]) and of level 2 (layers 23, 24: C = 24), forward and reverse, at sizes that span several
                           tiles of the fused kernels (B = 2, 36 x 40 / 18 x 20) -- the inputs the engine's DEFAULT hot path needs: hoist
                           producers (conv_up2_h2t / the 320 -> 1024 hoists, the 1x1-only head, conv_h2r) -> coupling_head -> coupling_tail
                           (FlowStep.py:88-129, FlowAffineCouplingsAblation.py:57-135).  The level-1 conditional is stored as its two parts
                           (key at 36 x 40, taps at 18 x 20): the reference is fed cat[key, F.interpolate(taps, 2, 'nearest')], which is what
                           SRFlowNet_arch.py:122-137 builds.
Usage:  python tests/golden/make_golden_steps_fused.py
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

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
    B = 2
    for tag, (la, lb), C, (H, W) in (("l1", (3, 4), 12, (36, 40)), ("l2", (23, 24), 24, (18, 20))):
        z = rnd(500 + la, B, C, H, W)
        if tag == "l1":
            key, taps = rnd(510, B, 64, H, W, scale=0.5), rnd(511, B, 256, H // 2, W // 2, scale=0.5)
            ft = torch.cat([key, F.interpolate(taps, scale_factor=2, mode="nearest")], 1)
            out.update(l1_key=key, l1_taps=taps)
        else:
            ft = rnd(520, B, 320, H, W, scale=0.5)
            out["l2_ft"] = ft
        ma, mb = net.flowUpsamplerNet.layers[la], net.flowUpsamplerNet.layers[lb]
        ld = torch.zeros(B)
        fa, _ = ma(z, ld, reverse=False, rrdbResults=ft)                 # encode order: a then b
        fab, _ = mb(fa, ld, reverse=False, rrdbResults=ft)
        rb, _ = mb(z, ld, reverse=True, rrdbResults=ft)                  # decode order: b then a
        rba, _ = ma(rb, ld, reverse=True, rrdbResults=ft)
        out.update({tag + "_z": z, tag + "_fwd_a": fa, tag + "_fwd_ab": fab, tag + "_rev_b": rb, tag + "_rev_ba": rba})
        pa, pb = "flowUpsamplerNet.layers.%d" % la, "flowUpsamplerNet.layers.%d" % lb
        ofa = O.flow_step(z, ft, sd, pa, True, False)
        orb = O.flow_step(z, ft, sd, pb, True, True)
        man[tag] = max(maxdiff(ofa, fa), maxdiff(O.flow_step(ofa, ft, sd, pb, True, False), fab),
                       maxdiff(orb, rb), maxdiff(O.flow_step(orb, ft, sd, pa, True, True), rba))
    out["weights_seed"] = np.int64(1234)
    out["weights_sha256"] = np.frombuffer(synth.digest(sd).encode(), dtype=np.uint8)
    save("srflow_steps_fused.npz", **out)
    path = os.path.join(HERE, "MANIFEST.json")
    full = json.load(open(path))
    full.setdefault("srflow", {})["flowstep_pairs_fused"] = man
    json.dump(full, open(path, "w"), indent=1, sort_keys=True)
    print("oracle vs reference:", man)


if __name__ == "__main__":
    main()
