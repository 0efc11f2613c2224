#!/usr/bin/env python3
"""arap_deform -- command-line twin of the reference's executable (ARAP/deformation/src/main.cpp:162-241).

  python arap_deform.py RGB Mask Constraint Flow warped_RGB warped_Mask       (one frame)
  python arap_deform.py listfile                                              (one solve per line, 6 paths)

Same argument contract, same fixed schedule (numIter 19, nonLinearIter 8, linearIter 400, main.cpp:215-221),
same border pins, same outputs (.flo + two PNGs).  ARAP_PLAN may name the reference's arap_plan.t; it is then
checked by Opt_ProblemDefine (anything that is not the ARAP energy is rejected); unset = built-in energy.
The GPU is selected with HIP_VISIBLE_DEVICES where the reference used CUDA_VISIBLE_DEVICES.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def usage():
    print("Usage:\n")
    print("./arap_deform RGB Mask Constraint Flow warped_RGB warped_Mask\n")
    print("Mask and warp image using the provided optical flow field.\n")
    print("RGB \t\t [input]  path to an input RGB image (.png only)")
    print("Mask\t\t [input]  path to an input mask image (.png only) where 0 for object, 1 for background")
    print("Constraint \t [input]  path to list of constraints, text file")
    print("Flow \t\t [output] path to optical flow image with (.flo only)")
    print("warped_RGB \t [output] path to output warped image (.png), all intermediate directories must exist")
    print("warped_Mask \t [output] path to output warped mask (.png), all intermediate directories must exist")


def main(argv):
    from arap_flow_amd import opt, pipeline
    if len(argv) == 7:
        lines = [tuple(argv[1:7])]
    elif len(argv) == 2:
        lines = pipeline.read_list(argv[1])
    else:
        print("Invalid Input!")
        usage()
        return 1
    if not lines:
        print("No file to be processed")
        return 1
    state = opt.State()
    plan = os.environ.get("ARAP_PLAN")
    if plan is not None:
        print("Optimization plan at %s" % plan)
        if not os.path.exists(plan):
            print(" Not found! Please run export ARAP_PLAN=/path/to/plan.t or copy the file to the running "
                  "folder with name arap_plan.t")
            return 1
        pr = state.lib.Opt_ProblemDefine(state.handle, plan.encode(), b"gaussNewtonGPU")
        if not pr:
            return 1
        state.lib.Opt_ProblemDelete(state.handle, pr)
    pipeline.deform_list(state, lines, 19, 8, 400)
    state.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
