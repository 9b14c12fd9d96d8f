"""CPU: include/rrt_line.h (closed form used by the kernels) against the oracle's literal walk of
rrt.py:202-229 -- every ordered pair of a 32x32 grid plus long random segments up to 2048."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include "rrt_line.h"
int32_t orc_bresenham_cells(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t *out_xy, int32_t cap);
static int32_t buf[2 * 5000];
static long check(int x0, int y0, int x1, int y1, long *tot) {
    long bad = 0;
    int c = orc_bresenham_cells(x0, y0, x1, y1, buf, 5000);
    rrt_line_t l = rrt_line_setup(x0, y0, x1, y1);
    if (c != l.major + 1) return 1;
    for (int k = 0; k < c; k++) { int x, y; rrt_line_cell(&l, k, &x, &y); (*tot)++; if (x != buf[2*k] || y != buf[2*k+1]) bad++; }
    return bad;
}
int main(void) {
    long bad = 0, tot = 0; int N = 32;
    for (int a = 0; a < N; a++) for (int b = 0; b < N; b++) for (int c = 0; c < N; c++) for (int d = 0; d < N; d++) bad += check(a, b, c, d, &tot);
    srand(1);
    for (int t = 0; t < 60000; t++) {
        int x0 = rand() % 2048, y0 = rand() % 2048, x1 = rand() % 2048, y1 = rand() % 2048;
        if (t % 5 == 0) x1 = x0 + (rand() % 5 - 2);
        if (t % 7 == 0) y1 = y0 + (rand() % 5 - 2);
        if (x1 < 0) x1 = 0; if (x1 > 2047) x1 = 2047; if (y1 < 0) y1 = 0; if (y1 > 2047) y1 = 2047;
        bad += check(x0, y0, x1, y1, &tot);
    }
    bad += check(0, 0, 2047, 2047, &tot) + check(2047, 2047, 0, 0, &tot) + check(0, 2047, 2047, 0, &tot) + check(0, 0, 2047, 1, &tot) + check(0, 0, 1, 2047, &tot);
    printf("%ld %ld\n", tot, bad);
    return bad != 0;
}
'''


def test_closed_form_equals_literal_walk(tmp_path):
    src = tmp_path / "linecheck.c"
    src.write_text(SRC)
    exe = tmp_path / "linecheck"
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src),
                           os.path.join(ROOT, "oracle", "rrt_oracle.c"), "-lm"])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) > 30_000_000 and int(out[1]) == 0
