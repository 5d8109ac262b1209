from .synth import SynthOCT    # noqa: F401


def EyeSetGenerator(dbname='synth', **kw):
    """reference data/octgen.py:35 entry point.  Only the synthetic GOALS-shaped generator is built in (the disk/cv2/
    albumentations pipeline is out of scope); `dbname='goals'` maps to the same 5-class shape."""
    if dbname in ('synth', 'goals'):
        return SynthOCT(dbname=dbname, **kw)
    raise ValueError(f"--db={dbname!r}: only 'synth'/'goals' (synthetic 5-class 800x1100 B-scans) is available here")
