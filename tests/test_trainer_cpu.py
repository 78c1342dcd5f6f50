"""Host logic of the training loops (iefvad_amd.trainer) that needs no GPU: the label vectors of train/utils.py:5-58."""
import torch

from iefvad_amd import trainer


def test_prompt_text_and_batch_labels_follow_the_reference_cases():
    ucf = {"Normal": "normal", "Abuse": "abuse", "Arrest": "arrest", "Arson": "arson"}          # any size but 2, 7, 17: the UCF case
    pt = trainer.get_prompt_text(ucf)
    assert pt == ["normal", "abuse", "arrest", "arson"]
    lab = trainer.get_batch_label(["Arson", "Normal", "Unknown"], pt, ucf)
    assert lab.tolist() == [[0, 0, 0, 1], [1, 0, 0, 0], [0, 0, 0, 0]]                            # a label outside the map stays all zero
    msad = {"Normal": "normal", "Abnormal": "abnormal"}
    assert trainer.get_batch_label(["Normal", "Fire"], trainer.get_prompt_text(msad), msad).tolist() == [[1, 0], [0, 1]]
    shang = {f"c{i}": f"t{i}" for i in range(17)}
    assert trainer.get_batch_label(["normal", "Normal"], trainer.get_prompt_text(shang), shang).tolist() == [[1, 0], [0, 1]]
    xd = {"A": "normal", "B1": "fighting", "B2": "shooting", "B4": "riot", "B5": "abuse", "B6": "car accident", "G": "explosion"}
    lab = trainer.get_batch_label(["B1-0-0", "B2-G-0", "A"], trainer.get_prompt_text(xd), xd)
    assert lab.shape == (3, 7) and lab[0].tolist() == [0, 1, 0, 0, 0, 0, 0] and lab[1].tolist() == [0, 0, 1, 0, 0, 0, 1]
    assert lab[2].tolist() == [1, 0, 0, 0, 0, 0, 0] and lab.dtype == torch.float32
