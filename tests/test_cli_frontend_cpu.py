"""CPU: the front-end program (linear_amd/csrc/linear_filter_main.cpp) against a TEST DOUBLE of the library's device half (tests/stub_abi.cpp; reader and
writer are the real host code): the reference's command-line surface (src/args_parser.cpp), several read files, output naming, N calculator threads
with N contexts, output in file order, and the gap stream protocol across contexts (the output must not depend on --gpus)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
CSRC = os.path.join(ROOT, "linear_amd", "csrc")


@pytest.fixture(scope="module")
def cli():
    os.makedirs(BUILD, exist_ok=True)
    so, exe = os.path.join(BUILD, "libstub_linear_amd.so"), os.path.join(BUILD, "linear_filter_stub")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "stub_abi.cpp"), os.path.join(CSRC, "lnr_reader.cpp"),
                           os.path.join(CSRC, "lnr_output.cpp"), "-o", so, "-lz", "-lpthread"])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", os.path.join(CSRC, "linear_filter_main.cpp"), "-o", exe, so, "-Wl,-rpath," + BUILD, "-lpthread"])
    return exe


def write_inputs(d, n_files=1, n_reads=240, seed=5):
    rng = np.random.default_rng(seed)
    abc = np.frombuffer(b"ACGT", np.uint8)
    with open(d / "ref.fa", "wb") as f:
        f.write(b">chrA some text\n" + abc[rng.integers(0, 4, 6000)].tobytes() + b"\n>chrB\n" + abc[rng.integers(0, 4, 3000)].tobytes() + b"\n")
    paths = []
    for k in range(n_files):
        p = d / f"reads{k}.part.fa"
        with open(p, "wb") as f:
            for i in range(n_reads):
                L = int(rng.integers(10, 400))
                s = abc[rng.integers(0, 3, L)].tobytes()          # no 'T' at the start ...
                if k == 0 and i == 97:
                    s = b"T" + s                                  # ... but for read 97 of the first file: the stream "extends" there
                f.write(b">r%d_%d extra words\n" % (k, i) + s + b"\n")
        paths.append(str(p))
    return paths, str(d / "ref.fa")


def run(cli, args, cwd):
    return subprocess.run([cli] + args, cwd=str(cwd), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


def test_output_does_not_depend_on_gpus_or_block_size(cli, tmp_path):
    reads, ref = write_inputs(tmp_path)
    outs = {}
    for tag, extra in {"g1": ["--gpus", "1", "--block-reads", "1000"], "g1b7": ["--gpus", "1", "--block-reads", "7"], "g3b7": ["--gpus", "3", "--block-reads", "7"],
                       "g4b1": ["--gpus", "4", "--devices", "0,0,0,0", "--block-reads", "1"], "g2build": ["--gpus", "2", "--index-mode", "build", "--block-reads", "16"]}.items():
        p = run(cli, ["filter", reads[0], ref, "-t", "2", "-ot", "3", "-o", str(tmp_path / tag), "-g", "50"] + extra, tmp_path)
        assert p.returncode == 0, p.stderr.decode()
        outs[tag] = (open(tmp_path / (tag + ".sam"), "rb").read(), [l for l in open(tmp_path / (tag + ".apf"), "rb").read().split(b"\n") if l])
    base = outs["g1"]
    assert base[0].count(b"\n") > 200 and b"@SQ\tSN:chrA\tLN:6000" in base[0] and b"@PG\tID:M1-3\tPN:Linear\tCL:\n" in base[0]
    for tag, o in outs.items():
        assert o == base, tag
    # the state really matters in the double: -g 0 gives another text
    p = run(cli, [reads[0], ref, "-ot", "2", "-o", str(tmp_path / "g0"), "-g", "0"], tmp_path)
    assert p.returncode == 0 and open(tmp_path / "g0.sam", "rb").read() != base[0]


def test_command_line_surface_of_the_reference(cli, tmp_path):
    reads, ref = write_inputs(tmp_path, n_files=3, n_reads=40)
    # several read files need the x separator (E[02]); with it: one output per read file, named by the file's name up to its first '.'
    p = run(cli, ["filter", reads[0], reads[1], ref], tmp_path)
    assert p.returncode == 1 and b"E[02]" in p.stderr
    p = run(cli, ["filter", reads[0], reads[1], reads[2], "x", ref, "-ot", "3", "--thread", "3", "--block-reads=9"], tmp_path)
    assert p.returncode == 0, p.stderr.decode()
    for k in range(3):
        sam = open(tmp_path / f"reads{k}.sam", "rb").read()
        assert sam.startswith(b"@SQ") and (b"r%d_0 extra words\t" % k) in sam and (b"r%d_" % ((k + 1) % 3)) not in sam
        assert os.path.exists(tmp_path / f"reads{k}.apf")
    # -o: one output for all read files; default -ot 2 = .sam only; bare -g (= 1) and -dup; -rg / -sn in the header; `filter` word optional
    p = run(cli, [reads[0], reads[1], "x", ref, "-o", "all", "-g", "-dup", "1", "-rg", "grp1", "-sn", "smp"], tmp_path)
    assert p.returncode == 0, p.stderr.decode()
    sam = open(tmp_path / "all.sam", "rb").read()
    assert sam.count(b"@SQ\tSN:chrA") == 1 and b"r0_5 extra words\t" in sam and b"r1_5 extra words\t" in sam and b"@RG\tID:grp1\tSM:smp\n" in sam
    assert not os.path.exists(tmp_path / "all.apf")
    # errors as the reference reports them
    assert run(cli, ["filter", reads[0]], tmp_path).returncode == 0 and b"SYNOPSIS" in run(cli, ["filter", reads[0]], tmp_path).stderr      # < 2 arguments: the help text
    p = run(cli, ["filter", reads[0], ref, "-zz", "1"], tmp_path)
    assert p.returncode == 1 and b"illegal option" in p.stderr
    p = run(cli, ["filter", reads[0], ref, "-t", "x3"], tmp_path)
    assert p.returncode == 1 and b"integer" in p.stderr
    p = run(cli, ["filter", reads[0], ref, "-t"], tmp_path)
    assert p.returncode == 1 and b"requires an argument" in p.stderr                                 # (a trailing option is not dropped silently)
    p = run(cli, ["filter", str(tmp_path / "nope.fa"), ref], tmp_path)
    assert p.returncode == 1 and b"E[05]" in p.stderr
    p = run(cli, ["filter", reads[0], str(tmp_path / "nope.fa")], tmp_path)
    assert p.returncode == 1 and b"E[06]" in p.stderr
    for bad in (["-ot", "4"], ["-ss"], ["-p", "0"], ["-c", "0"], ["-f", "1"], ["-r"], ["-b", "0"]):
        p = run(cli, ["filter", reads[0], ref] + bad, tmp_path)
        assert p.returncode == 1 and b"not built" in p.stderr, bad
    p = run(cli, ["filter", reads[0], ref, "-o", "/nonexistent_dir/x"], tmp_path)
    assert p.returncode == 1 and b"can't write" in p.stderr
