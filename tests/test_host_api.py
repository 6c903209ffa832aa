"""The C++ host API (rust-triplets_amd/host/triplets_amd.hpp) through its check program.

CPU: the program builds, fails loudly without a GPU, and the vectors it hard-codes agree with the oracle.
GPU: the program passes.
"""
import os
import subprocess

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "rust-triplets_amd", "host")


def build():
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    return os.path.join(HOST, "host_check")


def test_check_program_vectors_agree_with_the_oracle():
    cfg = O.make_config(True, 0.35, True, 1024, (64,), 512)
    r = O.batch(*O.pack([b"plain prose stays\n12 34 56 78\n| a | b |\n|---|---|", b"123 456\n---",
                         b"1234 5678\n99"]), cfg)
    assert r.record_text(0) == b"plain prose stays\na b"
    assert list(r.dropped) == [0, 1, 1]
    cfg = O.make_config(False, 0.35, True, 4, (1,), 2)
    r = O.batch(*O.pack([b"one two three four five six seven"]), cfg)
    assert [r.chunk_text(c) for c in range(len(r.chunks))] == [b"one two three four", b"four five six seven", b"one two"]
    text = "  alpha\tbeta\u3000gamma \n".encode()
    assert [text[a:a + n] for a, n in O.tokenize(text)] == [b"alpha", b"beta", b"gamma"]


def test_check_program_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2
    assert "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_check_program_passes_on_the_gpu():
    exe = os.path.join(HOST, "host_check")
    assert os.path.exists(exe), "build rust-triplets_amd/host beforehand (make -C rust-triplets_amd/host)"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "host_check: ok" in p.stdout
