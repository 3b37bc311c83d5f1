"""rau_vqa_amd: MI355X-native Recurrent Answering Unit forward/backward."""
