"""The slicing arithmetic of k_decode_sections on the host (no GPU): tests/crc_slices_check.cpp, built with plain g++ against
bloomsearch_amd/csrc/crc_slices.h, cuts payloads the way the kernel does (slices counted from the end, decode_unit / decode_splits),
combines the slices' zero-initial checksums with the GF(2) shifts the kernel uses and compares with a bit-by-bit CRC-32C
(the checksum encodeFilterSection stores, file_format.go:343-384).  Sizes: around the 16 KiB unit, the 32-slice limit where the unit
widens (a section's arrivals are 32 flag bits), and a section of a few MB."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [1, 2, 7, 8, 9, 63, 64, 65, 4095, 16383, 16384, 16385, 32768, 32769, 49152, 50000, 360_028, 524_287, 524_288, 524_289, 1_048_575, 1_048_576, 1_048_577,
         1_048_581, 1_048_640, 1_052_672, 3_000_001]


def test_slices_of_a_payload_combine_into_its_crc32c(tmp_path):
    exe = tmp_path / "crc_slices_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "bloomsearch_amd", "csrc"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "crc_slices_check.cpp")], check=True, timeout=120)
    r = subprocess.run([str(exe)] + [str(s) for s in SIZES], capture_output=True, text=True, timeout=120)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(lines) == len(SIZES) and all(ln.endswith(" ok") for ln in lines), r.stdout
    # the unit only widens once 32 slices of 16 KiB no longer cover the payload
    units = {int(ln.split()[0]): (int(ln.split()[1]), int(ln.split()[2])) for ln in lines}
    assert units[524_288] == (16384, 32) and units[524_289][0] > 16384 and units[524_289][1] <= 32
    assert all(n <= 32 for _, n in units.values())
    assert units[16384] == (16384, 1) and units[16385] == (16384, 2)
