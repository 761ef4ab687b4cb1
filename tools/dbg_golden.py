import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
from conftest import load_golden
from urh_amd.pipeline import DemodParams, DevicePipeline
pipe = DevicePipeline()
for name in ["homematic_i16","two_participants_i8"]:
    g = load_golden(name)
    p = DemodParams(g["modulation_type"], g["bits_per_symbol"], g["noise_threshold"], g["center"], g["center_spacing"],
                    g["tolerance"], g["samples_per_symbol"], g["costas_loop_bandwidth"], g["pause_threshold"], True)
    iq = torch.from_numpy(g["iq"]).cuda()
    res = pipe.iq_to_bits(iq, p, want_qad=True)
    q = res.qad.cpu().numpy(); w = g["qad"]
    bad = np.nonzero(q.view(np.uint32) != w.view(np.uint32))[0]
    print(name, g["modulation_type"], g["noise_threshold"], len(q), "bad", len(bad), bad[:20], bad[:20] % 8192, bad[:20] % 128)
    for b in bad[:10]:
        print(b, q[b], w[b], g["iq"][b-1:b+1].tolist())
    print("ppseq eq", np.array_equal(res.ppseq(), g["ppseq"]))
