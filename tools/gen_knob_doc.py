#!/usr/bin/env python3
"""Writes KNOBS.md from the library's own knob table (csrc/tor_knobs.hpp via tor_knob_count / tor_knob_info).

    python tools/gen_knob_doc.py            # rewrite KNOBS.md
    python tools/gen_knob_doc.py --check    # exit 1 when KNOBS.md is not what the table says (tests/test_knobs.py)
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def render() -> str:
    os.environ.setdefault("TOR_NO_TORCH", "1")
    tor = importlib.import_module("trace-of-radiance_amd")
    rows = tor.knobs()
    out = ["# Environment knobs of libtor_mi355x.so",
           "",
           "Generated from `trace-of-radiance_amd/csrc/tor_knobs.hpp` by `tools/gen_knob_doc.py` -- edit the table, not this file.",
           "Every `getenv` of the library goes through that table (`tests/test_knobs.py`).  **when**: `call` = read at every",
           "`tor_render*` call, `context` = read once at `tor_context_create` (the drop-in's cached contexts: the first use of a",
           "device in the process), `upload` = read when a scene's culling layout is built.  No knob changes a pixel except",
           "`TOR_DEFAULT_SEEDING` (a different, equally valid sample set); the TEST settings exercise failure paths and still",
           "return the right canvas.",
           "",
           "| knob | default | values | when | what |",
           "|---|---|---|---|---|"]
    for r in rows:
        what = r["what"].replace("|", "\\|")
        rng = r["range"].replace("|", "\\|")
        out.append(f"| `{r['name']}` | {r['default']} | {rng} | {r['when']} | {what} |")
    out += ["",
            "Harness-side switches (not in the library): `TOR_NO_TORCH=1` (Python package: do not import torch before loading the",
            "extension), `TOR_BENCH_BACKEND=gloo` (`bench.py`: several ranks on one GPU), `TOR_BENCH_RCCL_INIT_S` / `TOR_BENCH_RCCL_CHECK_S`",
            "(`bench.py`: deadlines of the library communicator's creation and of its self-check gather, default 120 / 60 s).", ""]
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "KNOBS.md")
    text = render()
    if "--check" in sys.argv:
        cur = open(path).read() if os.path.exists(path) else ""
        if cur != text:
            print("KNOBS.md is out of date: run python tools/gen_knob_doc.py")
            sys.exit(1)
        return
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path)


if __name__ == "__main__":
    main()
