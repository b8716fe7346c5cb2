"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side ("run CPU restatement under
ASan/UBSan"): the oracle (oracle/aai_oracle.c) and the host replay of the kernels' arithmetic together with the
product's planner (tests/emulation/host_emulation.cpp + csrc/aai_plan.cpp, aai_rot_math.hpp, aai_rot_quad.hpp,
aai_strict.hpp) are rebuilt with -fsanitize=address,undefined and replay the small and knife-edge golden cases.
GPU sanitizers are not available on the pool, so device code is covered through this shared-header replay only."""
import os
import subprocess
import sys

from conftest import BUILD, ROOT


def test_cpu_side_is_clean_under_asan_and_ubsan():
    os.makedirs(BUILD, exist_ok=True)
    flags = ["-O1", "-g", "-fPIC", "-shared", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
             "-ffp-contract=off"]
    oracle_so = os.path.join(BUILD, "liboracle_asan.so")
    emu_so = os.path.join(BUILD, "libaai_hostemu_asan.so")
    r = subprocess.run(["gcc", "-std=c11"] + flags + ["-o", oracle_so, os.path.join(ROOT, "oracle", "aai_oracle.c"), "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["g++", "-std=c++17"] + flags + ["-o", emu_so, os.path.join(ROOT, "tests", "emulation", "host_emulation.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(asan), "AddressSanitizer runtime not found"
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_replay.py"), oracle_so, emu_so],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "clean" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
