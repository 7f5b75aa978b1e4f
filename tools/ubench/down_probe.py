import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, cases, gnuspeech_amd as g
fr = cases.config2_frames(4096, nframes=251)
for rate in (16000.0, 8000.0):
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(rate)))
    st = b.prepare_device(fr)
    for _ in range(6): b.synthesize_device(st)
    torch.cuda.synchronize()
