# measures the file / WAV-data int16 differences test_synthesizer_facade_and_writers bounds
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, golden_io, oracle_lib as O, gnuspeech_amd as g
gold = golden_io.load("gnuspeech_window_44k")
pd = dict(gold["params_dict"])
pd2 = dict(pd, channels=2, balance=0.3)
op = O.InputParams.from_dict(pd2)
o = O.synthesize(op, gold["frames"].astype(np.float32).astype(np.float64))
b = g.TRMBatch(g.TRMInputParameters.from_dict(pd2))
pcm, ns, mx = b.synthesize([gold["frames"]])
e = (pcm[0].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
print("fp32 path: rms %.3e max-abs %.3e (of max); max diff %.3e rel" % (np.sqrt(np.mean(e*e)), np.abs(e).max(), abs(float(mx[0]) - o["maximumSampleValue"]) / o["maximumSampleValue"]))
for wav in (True, False):
    ref = O.scale_int16(op, o["samples"], o["maximumSampleValue"], for_wav_data=wav).astype(np.int32)
    own = O.scale_int16(op, pcm[0].astype(np.float64), float(mx[0]), for_wav_data=wav).astype(np.int32)
    d = ((own - ref + 32768) % 65536) - 32768
    print("for_wav_data=%s: max |int16 diff| %d, mean %.3f; predicted bound = max-abs error x 32767 x gain = %.2f" % (wav, np.abs(d).max(), np.abs(d).mean(), np.abs(e).max() * 32767 * (0.65 if wav else 1.3) + 32767*(0.65 if wav else 1.3)*abs(float(mx[0]) - o["maximumSampleValue"]) / o["maximumSampleValue"] + 0.5))
