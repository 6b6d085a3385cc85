"""DBoW2 transform at the reference's vocabulary size (k = 10, L = 6, 10^6 words), loaded through the text format:
load time, microseconds per transform of a frame's features (features resident on the device), per 1000 features.
   python3 tools/vocab_time.py [reps]          (under rocprofv3 --pmc FETCH_SIZE: traffic of vocab_transform_kernel per launch)
Algorithmic bytes per feature: its 32-byte descriptor + L levels x k children x 32 bytes of node descriptors + 16 bytes out."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                          # noqa: E402
from multi_orbslam3_amd import api, synth, views           # noqa: E402
from oracle import binding as ob                            # noqa: E402  (only the WRITER of the text file: test infrastructure)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    k, L = 10, 6
    v = synth.make_full_vocabulary(k, L)
    vv, keep = views.vocab_view(v["child_start"], v["child_ids"], v["desc"], v["weight"], v["word_id"], L)
    path = os.path.join(tempfile.mkdtemp(prefix="orbvoc_"), "voc.txt")
    ob.vocab_save_text(vv, k, path)
    size = os.path.getsize(path)
    t0 = time.perf_counter()
    arr = api.load_text_vocabulary(path)
    t_parse = time.perf_counter() - t0
    t0 = time.perf_counter()
    voc = api.ORBVocabulary.loadFromTextFile(path)
    t_load = time.perf_counter() - t0
    os.remove(path)
    sc = synth.Scene(640, 480, tex_size=(1600, 1200), px_per_m=200.0)
    Lm, Rm, Tcw = sc.stereo_pair(3)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
    (kl, dl), (kr, dr) = ex.extract_stereo(Lm, Rm)
    desc = np.concatenate([dl, dr])
    n = len(desc)
    out = {}
    for name, run in (("host descriptors (upload + walk + download)", lambda: voc.transform_features(desc, 4)),):
        for _ in range(10):
            run()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        us = 1e6 * (time.perf_counter() - t0) / reps
        out[name] = {"us_per_call": round(us, 1), "us_per_1000_features": round(us * 1000.0 / n, 1)}
    t0 = time.perf_counter()
    for _ in range(reps):
        voc.transform(desc, 4)
    us = 1e6 * (time.perf_counter() - t0) / reps
    out["transform() incl. BowVector / FeatureVector assembly"] = {"us_per_call": round(us, 1), "us_per_1000_features": round(us * 1000.0 / n, 1)}
    alg = n * (32 + L * k * 32 + 16)
    print(json.dumps({"vocabulary": {"k": k, "L": L, "nodes": int(len(arr["weight"])), "words": int(arr["n_words"]), "text_MB": round(size / 1e6, 1),
                                     "parse_s": round(t_parse, 3), "parse_plus_upload_s": round(t_load, 3)},
                      "features": n, "algorithmic_bytes_per_launch": alg, "timing": out}))


if __name__ == "__main__":
    main()
