"""Stub of OpenCV for importing the reference's common/utils.py (VideoRecorder, never used on this path)."""


def __getattr__(name):
    raise AttributeError("cv2 stub: %s is not available (video recording is outside the hot path)" % name)
