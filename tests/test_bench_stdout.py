"""bench.py's contract with the harness: rank 0 prints ONE JSON line on stdout.  Library banners (RCCL / gloo print theirs to file
descriptor 1 when a communicator comes up) and children must not share it: bench.claim_stdout() moves descriptor 1 to stderr and
emit_line() writes the line to the descriptor that was stdout.  CPU only (no torch, no device)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stdout_carries_the_json_line_and_nothing_else():
    code = ("import os, sys, subprocess; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "print('python-level chatter'); os.write(1, b'descriptor-level chatter (what a C library prints)\\n'); "
            "subprocess.run([sys.executable, '-c', 'print(\"a child\")']); bench.emit_line('{\"metric\": \"x\", \"value\": 1}')") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": "x", "value": 1}\n', r.stdout
    for noise in ("python-level chatter", "descriptor-level chatter", "a child"):
        assert noise in r.stderr


def test_emit_line_without_the_claim_is_a_plain_print():
    code = "import sys; sys.path.insert(0, %r); import bench; bench.emit_line('{}')" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout == "{}\n", (r.stdout, r.stderr)
