"""module path of the reference's LM wrappers (streammind/model/language_model/)"""
