from cutie_amd.inference.data.vos_test_dataset import VOSTestDataset  # noqa: F401
