#!/usr/bin/env python3
"""Writes the generated instruction streams (csrc/*_stream.inc) that changed: every generator renders into a temporary file and
an output is replaced only when its text differs, so an edit of one generator (or of the shared emitter) recompiles only the
translation units whose stream really changed.  Called by csrc/Makefile; usage: python tools/gen_streams.py <csrc directory>"""
import os
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

JOBS = [("p4gen", "write_inc", "attn_fwd16_p4_stream.inc"), ("p4pgen", "write_inc", "attn_fwd16_p4p_stream.inc"),
        ("dq4gen", "write_inc", "attn_dq16_p4_stream.inc"), ("dkv4gen", "write_inc", "attn_dkv16_p4_stream.inc"),
        ("f256gen", "write_inc", "attn_fwd16_p5_stream.inc"), ("f256gen", "write_one_operand_inc", "attn_fwd16_p5_tr1_stream.inc"),
        ("dkv5gen", "write_inc", "attn_dkv16_p5_stream.inc"), ("dq5gen", "write_inc", "attn_dq16_p5_stream.inc"),
        ("p6gen", "write_inc", "attn_fwd16_p6_stream.inc")]


def render(job):
    module, fn, name = job
    mod = __import__(module)
    with tempfile.NamedTemporaryFile("r", suffix=".inc", delete=False) as tmp:
        path = tmp.name
    getattr(mod, fn)(path)
    text = open(path).read()
    os.remove(path)
    return name, text


def main():
    csrc = sys.argv[1]
    with ProcessPoolExecutor(max_workers=min(8, len(JOBS))) as pool:
        for name, text in pool.map(render, JOBS):
            out = os.path.join(csrc, name)
            old = open(out).read() if os.path.exists(out) else None
            if old != text:
                with open(out, "w") as f:
                    f.write(text)
                print("gen_streams: wrote", name)


if __name__ == "__main__":
    main()
