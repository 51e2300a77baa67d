from cutie_amd.inference.utils.results_utils import ResultSaver, make_zip, davis_palette, davis_palette_np  # noqa: F401
