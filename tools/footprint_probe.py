"""resident bytes of a csa_wt by part, before and after sdsl_hip_fm_set_footprint, against the reference's stream size"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdsl-lite_amd")
for mib in [int(x) for x in sys.argv[1:]] or [3, 64]:
    text = pkg.english_text(mib << 20, 21)
    csa = pkg.csa_wt(text=text)
    blob = len(csa.serialize(32, 64, pkg.capi.LAYOUT_BV_MCL))
    print(mib, "MiB: sdsl stream", blob, "full", csa.device_bytes(), csa.footprint_parts(), "jump k", csa.jump_depth(), "kmer k", csa.kmer_table_depth())
    try:
        csa.set_footprint(int(1.5 * blob))
    except Exception as e:
        print("  1.5x:", e)
    print("  after 1.5x:", csa.device_bytes(), round(csa.device_bytes() / blob, 3), csa.footprint_parts(), "jump k", csa.jump_depth(), "kmer k", csa.kmer_table_depth())
    csa.close()
