#!/usr/bin/env python3
"""softwareTRM equivalent (Frameworks/Tube/main.m:12-67): `softwaretrm.py [-v] inputFile outputFile`.
Parses a .trm / Monet.parameters file, synthesizes on the GPU, writes the AU/AIFF/WAVE file."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnuspeech_amd as g  # noqa: E402


def main(argv):
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
    if verbose:
        print("input file:\t\t%s\n" % inp)
        d = tube.derived()
        print("actual tube length:\t%.4f cm\ninternal sample rate:\t%d Hz\ncontrol period:\t\t%d samples" % (
            d["actualTubeLength"], d["sampleRate"], d["controlPeriod"]))
        print("\nCalculating floating point samples...\nStarting synthesis")
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
