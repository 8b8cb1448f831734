#!/usr/bin/env python3
"""Regenerates the numeric fields of rng_kat.json from a from-scratch pure-Python restatement of
utopian/shaders/include/random.glsl:5-34 (integer arithmetic mod 2^32; the float conversion is
float32(word) / float32(4294967295.0), and float32(4294967295.0) == 2^32). The reference itself
cannot run in this environment; this script is the committed generator the
fixtures were checked against."""
import json
import os
import struct

M = 0xFFFFFFFF


def jenkins(x):
    x = (x + (x << 10)) & M
    x ^= x >> 6
    x = (x + (x << 3)) & M
    x ^= x >> 11
    x = (x + (x << 15)) & M
    return x


def init_rng(px, py, width, frame):
    return jenkins((px + py * width) ^ jenkins(frame))


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def random_float(state):
    state = (state * 747796405 + 1) & M
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & M
    word = (word >> 22) ^ word
    return f32(f32(float(word)) / 4294967296.0), state


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rng_kat.json")
    kat = json.load(open(path))
    kat["jenkins_hash"] = {k: jenkins(int(k)) for k in kat["jenkins_hash"]}
    for e in kat["init_rng"]:
        s = init_rng(e["px"], e["py"], e["width"], e["frame"])
        e["seed"] = s
        fl = []
        for _ in range(3):
            v, s = random_float(s)
            fl.append(round(v, 9))
        e["floats"] = fl
        if "state_after" in e:
            e["state_after"] = s
    json.dump(kat, open(path, "w"), indent=1)
    print(json.dumps(kat, indent=1))
