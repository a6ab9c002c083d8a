#!/usr/bin/env python3
"""Python-3 restatement of the reference results checker (reference check/check.py:1-147).

Same command line, same arithmetic, same messages and same exit codes as the reference's
Python-2.7-only script: compares av_vels.dat (column 1) and final_state.dat (column 5,
pressure) against reference files and fails when the largest per-element percentage
difference exceeds --tolerance (default 1 %) or is not finite.

    python check/check.py --ref-av-vels-file=R1 --ref-final-state-file=R2 \
                          --av-vels-file=A --final-state-file=F [--tolerance T]

`run_check()` is the importable form used by the tests; `main()` is the CLI.
"""
import argparse
import sys

import numpy as np


# (flag, help text) of the four file arguments, all required, one value each (check/check.py:32-57)
FILE_ARGUMENTS = (
    ("--ref-av-vels-file", "reference av_vels results file"),
    ("--ref-final-state-file", "reference final_state results file"),
    ("--av-vels-file", "calculated av_vels results file"),
    ("--final-state-file", "calculated final_state results file"),
)


def make_parser():
    """Argument set of the reference checker (check/check.py:17-57): same flags, defaults and help texts;
    arguments may also come from a file given as @file."""
    parser = argparse.ArgumentParser(description="Testing script for HPC LBM coursework", fromfile_prefix_chars="@",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--tolerance", nargs=1, type=float, default=[1],
                        help="Percentage tolerance to match against reference results")
    for flag, text in FILE_ARGUMENTS:
        parser.add_argument(flag, nargs=1, required=True, help=text)
    return parser


def load_dat_files(av_vels_filename, final_state_filename):
    """check/check.py:62-69: av_vels column 1; final_state columns x, y, pressure."""
    av_vels = np.loadtxt(av_vels_filename, usecols=[1])
    final_state = np.loadtxt(final_state_filename, usecols=[0, 1, 5])
    return np.atleast_1d(av_vels), np.atleast_2d(final_state)


def get_diff_values(ref_vals, sim_vals):
    """check/check.py:84-100: diff = ref - sim; percentage relative to (ref - diff) = sim."""
    diff = ref_vals - sim_vals
    with np.errstate(divide="ignore", invalid="ignore"):
        diff_pcnt = 100.0 * (diff / (ref_vals - diff))
    max_diff_step = int(np.argmax(np.abs(diff_pcnt)))
    return {
        "max_diff_step": max_diff_step,
        "max_diff": diff[max_diff_step],
        "max_diff_pcnt": diff_pcnt[max_diff_step],
        "sim_val": sim_vals[max_diff_step],
        "ref_val": ref_vals[max_diff_step],
        "total": np.sum(np.abs(diff)),
    }


AV_VELS_STRINGS = [
    "Total difference in av_vels : {total:.12E}",
    "Biggest difference (at step {max_diff_step:d}) : {max_diff:.12E}",
    "  {sim_val:.12E} vs. {ref_val:.12E} = {max_diff_pcnt:.2g}%",
]
FINAL_STATE_STRINGS = [
    "Total difference in final_state : {total:.12E}",
    "Biggest difference (at coord ({jj:d},{ii:d})) : {max_diff:.12E}",
    AV_VELS_STRINGS[2],
]


def run_check(ref_av_vels_file, ref_final_state_file, av_vels_file, final_state_file,
              tolerance=1.0, out=None):
    """Returns (exit_code, av_vels_diffs, final_state_diffs); prints the reference's messages."""
    out = out or sys.stdout

    def emit(s=""):
        print(s, file=out)

    av_vels_ref, final_state_ref = load_dat_files(ref_av_vels_file, ref_final_state_file)
    av_vels_sim, final_state_sim = load_dat_files(av_vels_file, final_state_file)

    # check/check.py:74-82
    if final_state_ref.shape != final_state_sim.shape or np.any(final_state_ref[:, 0:2] != final_state_sim[:, 0:2]):
        emit("Final state files coordinates were not the same")
        return 1, None, None
    if av_vels_ref.size != av_vels_sim.size:
        emit("Different number of steps in av_vels files")
        return 1, None, None

    av_vels_diffs = get_diff_values(av_vels_ref, av_vels_sim)
    for s in AV_VELS_STRINGS:
        emit(s.format(**av_vels_diffs))
    emit()

    final_state_diffs = get_diff_values(final_state_ref[:, 2], final_state_sim[:, 2])
    max_diff_loc = int(final_state_diffs["max_diff_step"])
    final_state_diffs["jj"] = int(final_state_sim[max_diff_loc, 0])
    final_state_diffs["ii"] = int(final_state_sim[max_diff_loc, 1])
    for s in FINAL_STATE_STRINGS:
        emit(s.format(**final_state_diffs))
    emit()

    # check/check.py:133-147
    final_state_failed = (not np.isfinite(final_state_diffs["max_diff_pcnt"])) or \
        (np.abs(final_state_diffs["max_diff_pcnt"]) > tolerance)
    av_vels_failed = (not np.isfinite(av_vels_diffs["max_diff_pcnt"])) or \
        (np.abs(av_vels_diffs["max_diff_pcnt"]) > tolerance)
    if final_state_failed:
        emit("final state failed check")
    if av_vels_failed:
        emit("av_vels failed check")
    if final_state_failed or av_vels_failed:
        return 1, av_vels_diffs, final_state_diffs
    emit("Both tests passed!")
    return 0, av_vels_diffs, final_state_diffs


def main(argv=None):
    args = make_parser().parse_args(argv)
    code, _, _ = run_check(args.ref_av_vels_file[0], args.ref_final_state_file[0],
                           args.av_vels_file[0], args.final_state_file[0], args.tolerance[0])
    return code


if __name__ == "__main__":
    sys.exit(main())
