"""CPU: the measurement / first-contact tooling must not rot -- every shell script parses, every Python tool compiles,
and the two round-5 HIP programs (the multi-GPU transport probe and the MFMA refresh probe) build for gfx950."""
import py_compile
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
TOOLS = ROOT / "tools"


@pytest.mark.parametrize("script", sorted(p.name for p in TOOLS.glob("*.sh")) + ["../jpeg-quantsmooth_amd/csrc/build_stripped.sh"])
def test_shell_scripts_parse(script):
    subprocess.run(["bash", "-n", str(TOOLS / script)], check=True)


@pytest.mark.parametrize("script", sorted(p.name for p in TOOLS.glob("*.py")))
def test_python_tools_compile(script, tmp_path):
    py_compile.compile(str(TOOLS / script), cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("src", ["first_contact_p2p.hip", "ubench_mfma_idct.hip"])
def test_round5_hip_programs_build(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-ffp-contract=off", str(TOOLS / src), "-o", str(tmp_path / "a.out")],
                   check=True, capture_output=True, timeout=600)


def test_first_contact_script_names_every_step():
    text = (TOOLS / "first_contact.sh").read_text()
    for piece in ("first_contact_p2p", "first_contact_shard.py", "bench.py --gpus", "tests/test_multigpu.py", "SUMMARY.txt", "--batch 1", "--quality 6"):
        assert piece in text, piece
