"""Writes tests/golden/known_answers.json: the published OpenCV input/output pairs quoted in
SURVEY.md Appendix A (A1 Lab, A2 HSV, A3 grey, A5 ellipse kernels) plus hand-derived morphology /
CCL cases whose expected values follow from the definitions alone.  No reference code is run (the
reference has no tests or fixtures and cv2 is absent); this script only records constants."""
import json
import os

known = {
    "source": "SURVEY.md Appendix A (widely published cv2 outputs); hand-derived cases marked 'by definition'",
    "bgr2lab": [  # (B,G,R) -> (L,a,b), cv2.cvtColor(COLOR_BGR2LAB) on uint8
        [[255, 255, 255], [255, 128, 128]], [[0, 0, 0], [0, 128, 128]], [[0, 0, 255], [136, 208, 195]],
        [[0, 255, 0], [224, 42, 211]], [[255, 0, 0], [82, 207, 20]], [[128, 128, 128], [137, 128, 128]]],
    "bgr2hsv": [  # (B,G,R) -> (H,S,V), COLOR_BGR2HSV uint8 (H in [0,180))
        [[0, 0, 255], [0, 255, 255]], [[0, 255, 0], [60, 255, 255]], [[255, 0, 0], [120, 255, 255]],
        [[255, 255, 255], [0, 0, 255]], [[0, 0, 0], [0, 0, 0]], [[0, 128, 255], [15, 255, 255]],
        [[255, 0, 128], [135, 255, 255]], [[140, 170, 200], [15, 76, 200]]],
    "bgr2gray": [[[255, 255, 255], 255], [[0, 0, 255], 76], [[0, 255, 0], 150], [[255, 0, 0], 29], [[0, 0, 0], 0]],
    "lab_coeffs": [1777, 1541, 778, 871, 2929, 296, 73, 448, 3575],
    "ellipse5": ["00100", "11111", "11111", "11111", "00100"],
    "ellipse7": ["0001000", "0111110", "1111111", "1111111", "1111111", "0111110", "0001000"],
    "ellipse3": ["010", "111", "010"],
    "inrange": {"values": [0, 1, 149, 150, 151, 254, 255], "lo": 150, "hi": 255, "expect": [0, 0, 0, 255, 255, 255, 255]},
    # by definition: 7x7 mask with a 3x3 block; erode 3x3 leaves its centre, dilate 3x3 grows it to 5x5,
    # the image border never wins (a full image stays full under erosion)
    "morph": {"block": [2, 2, 3, 3], "size": [7, 7]},
    # by definition (Appendix A8): two single pixels, A at (row 1, col 0), B at (row 0, col 10):
    # 2x2-block raster order labels A=1, B=2; pixel raster order labels B=1, A=2
    "ccl_numbering": {"size": [4, 16], "A": [1, 0], "B": [0, 10], "block2x2": {"A": 1, "B": 2}, "pixel": {"A": 2, "B": 1}},
    # (H,S,V) -> (B,G,R), COLOR_HSV2BGR uint8: the inverses of the A2 pairs plus two secondaries
    "hsv2bgr": [[[0, 255, 255], [0, 0, 255]], [[60, 255, 255], [0, 255, 0]], [[120, 255, 255], [255, 0, 0]], [[0, 0, 255], [255, 255, 255]],
                [[0, 0, 0], [0, 0, 0]], [[15, 76, 200], [140, 170, 200]], [[30, 255, 255], [0, 255, 255]], [[90, 255, 128], [128, 128, 0]]],
    # cv2.getGaussianKernel's small fixed kernels scaled by 256 (the 8.8 taps of the bit-exact 8-bit GaussianBlur)
    "gaussian_taps": {"3": [64, 128, 64], "5": [16, 64, 96, 64, 16], "7": [8, 28, 56, 72, 56, 28, 8], "9": [4, 13, 30, 51, 60, 51, 30, 13, 4]},
    # cv2.findContours on a filled 5x4 rectangle at (3,2): top-left, bottom-left, bottom-right, top-right (CHAIN_APPROX_SIMPLE)
    "contour_rect": {"size": [8, 10], "rect": [2, 3, 6, 8], "points": [[3, 2], [3, 5], [7, 5], [7, 2]]},
    # (B,G,R) -> (Y,Cr,Cb), COLOR_BGR2YCrCb uint8, and -> (H,L,S), COLOR_BGR2HLS uint8 (H in [0,180)): primaries, secondaries, greys
    "bgr2ycrcb": [[[0, 0, 255], [76, 255, 85]], [[255, 0, 0], [29, 107, 255]], [[0, 255, 0], [150, 21, 43]], [[255, 255, 255], [255, 128, 128]],
                  [[0, 0, 0], [0, 128, 128]], [[128, 128, 128], [128, 128, 128]]],
    "bgr2hls": [[[0, 0, 255], [0, 128, 255]], [[0, 255, 0], [60, 128, 255]], [[255, 0, 0], [120, 128, 255]], [[255, 255, 255], [0, 255, 0]],
                [[0, 0, 0], [0, 0, 0]], [[128, 128, 128], [0, 128, 0]], [[0, 255, 255], [30, 128, 255]], [[255, 255, 0], [90, 128, 255]],
                [[255, 0, 255], [150, 128, 255]]],
    # by definition: integer translation by cv2.warpAffine is a shift with the border value elsewhere
    "warp_translate": {"size": [6, 8], "shift": [3, -2], "border_value": 0},
    # cv2.threshold(THRESH_OTSU) on a two-valued image returns the lower value
    "otsu_two_values": {"values": [10, 200], "threshold": 10},
}

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json"), "w") as f:
    json.dump(known, f, indent=1)
