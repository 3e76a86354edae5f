"""The C-ABI library loads (no GPU needed) and exports every function include/svjg.h declares; without a GPU the
product path fails loudly instead of falling back to anything."""
import os
import re

import pytest


def _declared(root):
    txt = open(os.path.join(root, "include", "svjg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svjg_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(root):
    from svjg import capi
    lib, host = capi.load_library(), capi.load_host_library()
    names = _declared(root)
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n) or hasattr(host, n), f"neither libsvjg_hip.so nor libsvjg_host.so exports {n}"
    assert set(names) == set(capi.EXPORTS), "svjg/capi.py prototypes and include/svjg.h disagree"
    assert lib.svjg_abi_version() == 1


def test_no_cpu_fallback(root):
    from svjg import capi
    lib = capi.load_library()
    if lib.svjg_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.SvjgError) as ei:
        capi.Context(0)
    assert "no HIP device" in str(ei.value)
    # the drop-in script exits with code 1 (uncaught exception), like any failure of the reference's script
    import subprocess
    import sys
    q = os.path.join(root, "tests", "golden", "quirks")
    p = subprocess.run([sys.executable, os.path.join(root, "svjedi-graph_amd", "filter-alignments.py"), "-a", f"{q}/alt_del.gaf",
                        "-g", f"{q}/q.gfa", "-p", f"{q}/q"], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr
    assert not os.path.exists(f"{q}/q_informative_aln.json")


def test_product_never_imports_the_oracle(root):
    pkg = os.path.join(root, "svjedi-graph_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in src.lower() or f == "svjg_line.h", f"{f} mentions the oracle"
