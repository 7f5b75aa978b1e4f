#!/usr/bin/env python3
"""softwareTRM equivalent (Frameworks/Tube/main.m:12-67): `softwaretrm.py [-v] inputFile outputFile`.
Parses a .trm / Monet.parameters file, synthesizes on the GPU, writes the AU/AIFF/WAVE file.

Batch directory mode (no reference counterpart: the reference runs one file per process):
    softwaretrm.py --batch inputDir outputDir
synthesizes every *.trm / *.parameters file of inputDir; files that share their input parameters go to the
GPU as ONE launch (a batch of voices), and each gets its own sound file (same name, the extension of its
outputFileFormat) exactly as the single-file mode would have written it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnuspeech_amd as g  # noqa: E402


def batch_mode(indir, outdir):
    import ctypes as C
    import numpy as np
    os.makedirs(outdir, exist_ok=True)
    groups = {}
    for name in sorted(os.listdir(indir)):
        if not name.endswith((".trm", ".parameters")):
            continue
        data = g.TRMDataList.initWithContentsOfFile(os.path.join(indir, name))
        if data is None:
            sys.stderr.write("%s: cannot parse, skipped\n" % name)
            continue
        groups.setdefault(bytes(data.inputParameters.c), []).append((name, data))
    ext = {0: ".au", 1: ".aiff", 2: ".wav"}
    nfiles = 0
    for members in groups.values():
        ip = members[0][1].inputParameters
        batch = g.TRMBatch(ip)
        pcm, ns, mx = batch.synthesize([d.frame_array() for _, d in members])
        for (name, _), samples, n, m in zip(members, pcm, ns, mx):
            out = os.path.join(outdir, os.path.splitext(name)[0] + ext.get(ip.outputFileFormat, ".au"))
            a = np.ascontiguousarray(samples, dtype=np.float32)
            g._capi.check(g.lib().trm_write_sound_file(C.byref(ip.c), a.ctypes.data, int(n), float(m), out.encode()))
            nfiles += 1
    print("%d files in %d launches" % (nfiles, len(groups)))
    return 0


def main(argv):
    if len(argv) == 4 and argv[1] == "--batch":
        return batch_mode(argv[2], argv[3])
    verbose = False
    if len(argv) == 3:
        inp, out = argv[1], argv[2]
    elif len(argv) == 4 and argv[1] == "-v":
        verbose, inp, out = True, argv[2], argv[3]
    else:
        sys.stderr.write("Usage:  %s [-v] inputFile outputFile\n" % argv[0])
        return 255
    data = g.TRMDataList.initWithContentsOfFile(inp)
    if data is None:
        sys.stderr.write("Aborting...\n")
        return 255
    tube = g.TRMTubeModel.initWithInputData(data)
    if tube is None:
        sys.stderr.write("Aborting...\n")
        return 255
    if verbose:                                               # main.m:44-51
        print("input file:\t\t%s\n" % inp)
        tube.printInputData()
        print("\nCalculating floating point samples...\nStarting synthesis")
        sys.stdout.flush()
    tube.synthesize()
    if verbose:
        print("done.")
    if not tube.saveOutputToFile(out):
        sys.stderr.write("Failed to save output\n")
        return 1
    if verbose:
        print("\nWrote scaled samples to file:  %s" % out)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
