from cutie_amd.inference.data.video_reader import VideoReader  # noqa: F401
