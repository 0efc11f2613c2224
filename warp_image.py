#!/usr/bin/env python3
"""warp_image -- command-line twin of ARAP/warping/src/main.cpp:302-336.

  python warp_image.py image mask flow warped_image warped_mask
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def usage():
    print("Usage:")
    print("./warp_image image mask flow warped_image warped_mask")
    print("Mask and warp image using the provided optical flow field.")
    print("\timage: path to image with png extension")
    print("\tmask: path to mask image with png extension, 0 for object, 1 for background")
    print("\tflo: path to optical flow image with flo extension")
    print("\twarped_image: path to output warped image (.png), all intermediate directories must exist")
    print("\twarped_mask: path to output warped mask (.png), all intermediate directories must exist")


def main(argv):
    if len(argv) != 6:
        print("Invalid Input! ", end="")
        usage()
        return 1
    from arap_flow_amd import opt, pipeline
    state = opt.State()
    pipeline.warp_files(state, *argv[1:6])
    state.close()
    print("Saved")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
