"""CPU: the measurement tooling parses before any GPU minute is spent on it -- every tools/*.py compiles, every tools/*.sh
passes `bash -n`, and no shell script embeds Python in a heredoc (round-2 finding: a quoting error inside such a heredoc let
five rocprofv3 passes run and then died before writing their summary)."""
import glob
import os
import py_compile
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tool_scripts_parse():
    pys = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert pys
    for p in pys:
        py_compile.compile(p, doraise=True)
    shs = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert shs
    for p in shs:
        r = subprocess.run(["bash", "-n", p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)
        assert not re.search(r"python3?\s+-\s*<<", open(p).read()), f"{p}: Python heredoc -- move it into tools/pmc_post.py"


def test_pmc_post_names_kernel_families():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_post", os.path.join(ROOT, "tools", "pmc_post.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    n = "void (anonymous namespace)::conv3x3_pipe_kernel<5, false, 128, true>(ConvArgs)"
    assert m.short(n) == "conv3x3_pipe_kernel<5, false, 128, true>" and m.family(n) == "conv3x3"
    assert m.family("void (anonymous namespace)::wgrad_dma_kernel<4, 2, false>(WgradArgs)") == "wgrad3x3"
    assert m.family("_ZN12_GLOBAL__N_112wgrad_kernelIDF16bLi2ELi4ELi2EEEv9WgradArgs") == "wgrad_other"
    assert m.family("void at::native::vectorized_elementwise_kernel<4>(int)") is None
