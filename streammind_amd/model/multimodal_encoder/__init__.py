"""module path of the reference's vision tower (streammind/model/multimodal_encoder/)"""
