"""``python -m leopard_amd.run_reference_eval <script.py> -- <script arguments>``

Runs one of the reference's evaluation scripts (evaluations/models/llava_multiimg_siglip_anyres.py,
evaluations/models/idefics2_multiimg.py) as ``__main__``, unmodified and in place, with the model classes it imports from
``transformers`` bound to the HIP engine (leopard_amd.reference_shim).  The working directory is left alone: the scripts
read ``../eval_<dataset>.jsonl`` relative to it, exactly as under the reference's own launcher
(run_eval_llava_siglip_multiimg.sh:9-11)."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    rest = argv[1:]
    if rest and rest[0] == "--":
        rest = rest[1:]
    from . import reference_shim
    reference_shim.install()
    sys.argv = [script] + rest
    sys.path.insert(0, os.path.dirname(script))           # what `python script.py` does
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
