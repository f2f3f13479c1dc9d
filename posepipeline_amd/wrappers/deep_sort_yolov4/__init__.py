"""Drop-in for pose_pipeline/wrappers/deep_sort_yolov4/ (tracking_method 0, `DeepSortYOLOv4`)."""
