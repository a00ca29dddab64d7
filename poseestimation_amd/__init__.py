"""MI355X-native OpenPose decode (heatmaps + PAFs -> skeletons)."""
