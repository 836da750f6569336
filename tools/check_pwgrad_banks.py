#!/usr/bin/env python3
"""Exhaustive bank check of pwgrad_kernel's LDS swizzles (csrc/conv_wgrad.hip).  A ds_read_b64_tr_b16 is served half a wave at a
time: lanes 0-31 (g = 0, 1; q = 0..3) read 8 pixel rows x 32 bytes.  With 64 banks x 4 B the read is conflict-free iff those eight
32-byte pieces fall on eight distinct 32-byte groups modulo 256.  Checked for every tap offset, patch row, channel block and half."""


def x_conflicts(C, stride):
    XW, rowb = stride * 32 + 2, C * 2
    swz = (lambda l: (l >> 3) & 1) if C == 32 else (lambda l: ((l >> 1) & 1) | (((l >> 3) & 1) << 1))
    bad = 0
    for half in (0, 1):
        for off in range(0, 36):                 # tap column offset (stride 2: (tx & 1) * 33 + (tx >> 1))
            for py in range(10):
                for blk in range(C // 16):
                    for hi in (0, 4):
                        groups = set()
                        for g in (2 * half, 2 * half + 1):
                            for q in range(4):
                                lcol = off + 8 * g + q + hi
                                groups.add((((py * XW + lcol) * rowb + ((blk ^ swz(lcol)) << 5)) % 256) // 32)
                        bad += len(groups) != 8
    return bad


def y_conflicts():
    swz = lambda r: ((r >> 1) & 1) | (((r >> 3) & 1) << 1)
    bad = 0
    for half in (0, 1):
        for r in range(8):
            for blk in range(4):
                for hi in (0, 4):
                    groups = set()
                    for g in (2 * half, 2 * half + 1):
                        for q in range(4):
                            row = r * 32 + 8 * g + q + hi
                            groups.add(((row * 128 + ((blk ^ swz(row)) << 5)) % 256) // 32)
                    bad += len(groups) != 8
    return bad


if __name__ == '__main__':
    for C in (32, 64):
        for s in (1, 2):
            print(f'X  Cin={C} stride={s}: conflicting reads {x_conflicts(C, s)}')
    print(f'dY 64 channels: conflicting reads {y_conflicts()}')
