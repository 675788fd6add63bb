"""Data formats on the input side of the path (SURVEY 8(f) "next"): the reference's split-file / PNG dataset readers
(`dataloader/dataloader.py`, `dataloader/dataloaderSR.py`, `dataloader/data_util.py`) re-stated without OpenCV."""
