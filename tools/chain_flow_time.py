#!/usr/bin/env python3
"""Config-4 squeeze chain (+ RCT) at size^2: the dataflow launch of the streamed levels (k6_unsqueeze_flow, default)
against one launch per level (JXLH_CHAIN_FLOW=0), timed with the context's events, interleaved, warm."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jxl_rs_amd
from jxl_rs_amd.modular import ModularChain


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = jxl_rs_amd.Context(0, 1)
    out = {"size": n}
    for rct in ((6, 0), None):
        ch = ModularChain(ctx, n, n, seed=84, rct=rct)

        def timed(flow):
            os.environ["JXLH_CHAIN_FLOW"] = "1" if flow else "0"
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.08:
                ch.run_chain()
                ctx.sync()
            ctx.timer_start()
            for _ in range(reps):
                ch.run_chain()
            return round(ctx.timer_stop() / reps, 4)

        key = "with_rct" if rct else "no_rct"
        out[key] = {"flow_ms": [], "levels_ms": []}
        for _ in range(3):
            out[key]["flow_ms"].append(timed(True))
            out[key]["levels_ms"].append(timed(False))
        os.environ["JXLH_CHAIN_FLOW"] = "1"
        ctx.flow_profile(True)
        ch.run_chain()
        ctx.sync()
        ch.run_chain()
        ctx.sync()
        out[key]["levels"] = [(hz, ow, oh) for hz, ow, oh in ch.steps]
        tl = ctx.flow_profile(False)
        flow_levels = out[key]["levels"][len(out[key]["levels"]) - len(tl) - (1 if rct else 0):]
        for row, (hz, ow, oh) in zip(tl, flow_levels):
            wgs = 3 * (((oh if hz else ow) + 63) // 64)
            chunks = ((ow if hz else oh) + 63) // 64
            row["level"] = ("H" if hz else "V") + " %dx%d" % (ow, oh)
            row["us_per_chunk_unblocked"] = round((row["lifetime_us_sum"] - row["poll_wait_us_sum"]) / (wgs * chunks), 2)
        out[key]["flow_timeline"] = tl
        ch.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
