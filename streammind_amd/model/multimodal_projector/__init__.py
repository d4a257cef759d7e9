"""module path of the reference's projectors (streammind/model/multimodal_projector/)"""
