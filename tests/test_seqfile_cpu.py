"""CPU: the input side of `mcq` (SeqFile, mcq_common.h) -- the chunk-by-chunk streaming index must deliver exactly the records the
all-at-once index delivers, for every chunk size, including files that fall back to the exact sequential reader in mid-stream."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_streaming_index_equals_eager_index(tmp_path):
    exe = str(tmp_path / "seqfile_stream_check")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "metacache_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "seqfile_stream_check.cpp"), "-o", exe, "-lz", "-pthread"])
    files = [os.path.join(GOLD, f) for f in ("cli_reads.fa", "cli_pairs.fq", "cli_irregular.fq", "cli_irregular.fa", "cli_p1.fa", "cli_truth.fa", "cli_fmt.fa")]
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("records ok") == len(files)
