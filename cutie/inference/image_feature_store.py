from cutie_amd.inference.image_feature_store import ImageFeatureStore  # noqa: F401
