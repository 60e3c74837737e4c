#!/usr/bin/env python3
"""Exhaustive check of the ITERATION PIPELINE of the ROF box sweeps (optical-flow-1_amd/csrc/ofx_occ.hip, k_rof_window with
several iterations of Scalar_ROF_BoxCellCentered in flight).

Iteration s of the solver is: alfa_s = |grad u_{s-1}| / (lambda g) per cell, one in-place sweep over the dual pairs PP with
alfa_s, u_s = lambda f + lambda div PP.  The sweep of iteration s runs LAGI steps behind the sweep of iteration s - 1; between
the two wavefronts an "alfa stage" computes alfa_s (through u_{s-1}) from the values the sweep s - 1 has left behind, D steps
ahead of the sweep s.  With K steps per launch, row blocks of R rows LAG steps apart:

  sweep  s, cell Z      : launch  Ls(s, Z) = floor((pos(Z)     + LAG blk(Z) + LAGI s) / K),      pos = 2 ci + cj
  alfa   s, cell X      : launch  La(s, X) = floor((pos(X) - D + LAG blk(X) + LAGI s) / K)        (s >= 1; alfa_0 is a full pass)

Kernel boundaries are the only synchronisation between workgroups, so for every cell:
  R1  the alfa stage s reads PP[Y], PP[N(Y)].x, PP[W(Y)].y for Y in {X, E(X), S(X)}: each must hold the FINAL value of sweep
      s - 1, stored by an earlier launch, and sweep s must first touch it in a later launch;
  R2  sweep s at Z reads alfa_s of W, N, Z, E, NE: written by an earlier launch;
  R3  alfa stage s overwrites alfa(C) only after every cell of sweep s - 1 that reads it (C, E, S, W, SW of C) has run in an
      earlier launch;
  R4  sweep s at Z reads the pairs of its ten stencil cells: those sweep s has not updated yet hold sweep s - 1's final values
      stored by an earlier launch; and sweep s first writes a pair only after sweep s - 1's last reader of it has run.
violations(...) returns the number of violated (cell, condition) pairs; run as a script for a table."""
import sys


def violations(nx, ny, R, K, LAG, LAGI, D, n_iter=3, verbose=False):
    pos = lambda c: 2 * c[0] + c[1]
    blk = lambda c: c[0] // R
    inside = lambda c: 0 <= c[0] < ny and 0 <= c[1] < nx
    Ls = lambda s, c: (pos(c) + LAG * blk(c) + LAGI * s) // K
    La = lambda s, c: (pos(c) - D + LAG * blk(c) + LAGI * s) // K
    N = lambda c: (c[0] - 1, c[1]); S = lambda c: (c[0] + 1, c[1]); W = lambda c: (c[0], c[1] - 1); E = lambda c: (c[0], c[1] + 1)

    def last_write(s, c, comp):                      # launch of the last store of sweep s into PP[c].x (0) / .y (1)
        o = S(c) if comp == 0 else E(c)
        return Ls(s, o) if inside(o) else Ls(s, c)

    def readers(c):                                  # cells whose stencil contains the pair of c (rof_cell's ten reads)
        ci, cj = c
        return [z for z in [(ci, cj + 2), (ci, cj + 1), (ci, cj), (ci, cj - 1), (ci + 1, cj + 1), (ci + 1, cj), (ci + 1, cj - 1),
                            (ci + 2, cj), (ci - 1, cj), (ci - 1, cj + 1)] if inside(z)]

    bad = 0
    def fail(tag, *a):
        nonlocal bad
        bad += 1
        if verbose and bad < 6:
            print("viol", tag, a)
    cells = [(i, j) for i in range(ny) for j in range(nx)]
    for s in range(1, n_iter):
        for X in cells:
            la = La(s, X)
            if la < 0:
                fail("alfa launch negative", s, X)
            for Y in (X, E(X), S(X)):                                   # R1
                if not inside(Y):
                    continue
                for e, comp in ((Y, 0), (Y, 1), (N(Y), 0), (W(Y), 1)):
                    if not inside(e):
                        continue
                    if not last_write(s - 1, e, comp) < la:
                        fail("R1 final", s, X, e, comp)
                    if not la < Ls(s, e):
                        fail("R1 untouched", s, X, e, comp)
            for Z in (X, E(X), S(X), W(X), (X[0] + 1, X[1] - 1)):      # R3: readers of alfa(X) in sweep s - 1
                if inside(Z) and not Ls(s - 1, Z) < la:
                    fail("R3", s, X, Z)
        for Z in cells:
            ls = Ls(s, Z)
            for C in (W(Z), N(Z), Z, E(Z), (Z[0] - 1, Z[1] + 1)):      # R2
                if inside(C) and not La(s, C) < ls:
                    fail("R2", s, Z, C)
            ci, cj = Z
            for e in [(ci, cj - 2), (ci, cj - 1), (ci, cj), (ci, cj + 1), (ci - 1, cj - 1), (ci - 1, cj), (ci - 1, cj + 1),
                      (ci - 2, cj), (ci + 1, cj), (ci + 1, cj - 1)]:   # R4
                if not inside(e):
                    continue
                for comp in (0, 1):
                    if not last_write(s - 1, e, comp) < ls:
                        fail("R4 final", s, Z, e, comp)
            for Zp in readers(Z):
                if not Ls(s - 1, Zp) < ls:
                    fail("R4 war", s, Z, Zp)
    return bad


if __name__ == "__main__":
    K, LAG = 24, 32
    for LAGI, D in [(128, 58), (120, 58), (112, 50), (128, 48), (144, 64), (160, 72)]:
        tot = 0
        for nx, ny, R in [(40, 30, 125), (17, 300, 125), (64, 260, 125), (9, 9, 125), (2, 2, 125), (33, 20, 7), (20, 64, 16), (5, 40, 3)]:
            tot += violations(nx, ny, R, K, LAG, LAGI, D)
        print("K", K, "LAG", LAG, "LAGI", LAGI, "D", D, "violations", tot)
